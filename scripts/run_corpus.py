#!/usr/bin/env python3
"""BASELINE.json config 5 for real: the 10 h corpus through flac_amd.corpus (on as many ranks as torchrun gives it),
then -- outside the product path -- the checks SURVEY.md 8d lists for it: the first >= 10 minutes of the stream
byte-identical to the unmodified reference's encoding of the same samples, every frame's CRC-16 (done by the job
itself), STREAMINFO's MD5 against hashlib over the regenerated input.  Writes gpurun_out/<tag>/corpus.json.
usage: python scripts/run_corpus.py [--tag T] [--hours H] [--gather rccl|host] [--nproc N]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="corpus")
    ap.add_argument("--hours", type=float, default=10.0)
    ap.add_argument("--gather", default="rccl")
    ap.add_argument("--nproc", type=int, default=1)
    ap.add_argument("--prefix-minutes", type=float, default=10.0)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    outdir = os.path.join(ROOT, "gpurun_out", args.tag)
    os.makedirs(outdir, exist_ok=True)
    flac = "/tmp/corpus_%s.flac" % args.tag
    cmd = [sys.executable]
    if args.nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.nproc), "--master-addr", "127.0.0.1", "--master-port", "29547"]
    cmd += ["-m", "flac_amd.corpus", "--hours", str(args.hours), "--gather", args.gather, "--out", flac]
    if args.nproc == 1:
        cmd += ["--force-dist"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29546"))
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit("corpus job failed")
    line = json.loads(r.stdout.strip().splitlines()[-1])
    line["job_wall_seconds_incl_startup_md5_write"] = round(wall, 2)

    from flac_amd import corpus as co
    from oracle import pyoracle as po
    base = co.base_clip()
    nfr = int(args.prefix_minutes * 60 * co.RATE) // co.BLOCK + 1
    pcm = co.host_frames(base, 0, nfr).astype(np.int32)
    t0 = time.perf_counter()
    ref = po.ref_encode(pcm, 16, co.RATE, 8) if po.have_ref() else po.oracle_encode(pcm, 16, co.RATE, 8)
    ref_frames = ref["data"][ref.get("header_bytes", 0):]
    t_ref = time.perf_counter() - t0
    with open(flac, "rb") as f:
        head = f.read(line["header_bytes"])
        got = f.read(len(ref_frames))
    line["prefix_check"] = {"frames": nfr, "minutes": round(nfr * co.BLOCK / co.RATE / 60, 2), "bytes": len(ref_frames),
                            "identical_to": "reference libFLAC (oracle/_ref)" if po.have_ref() else "oracle",
                            "identical": got == ref_frames, "reference_encode_seconds": round(t_ref, 2),
                            "reference_Msamples_per_s_1thread": round(nfr * co.BLOCK / t_ref / 1e6, 2)}
    # STREAMINFO as patched at finish: MD5 field == hashlib over the regenerated input, total samples, frame size bounds
    si = head[8:8 + 34]
    h = hashlib.md5()
    total = line["samples"]
    F = line["frames"]
    for f0 in range(0, F, 8192):
        x = co.host_frames(base, f0, min(F, f0 + 8192))
        h.update(x[:max(0, min(x.shape[0], total - f0 * co.BLOCK))].tobytes())
    line["streaminfo_check"] = {"md5_field": si[18:34].hex(), "md5_of_input": h.hexdigest(), "md5_ok": si[18:34] == h.digest(),
                                "total_samples_field": int.from_bytes(si[10:18], "big") & ((1 << 36) - 1), "total_samples_ok": (int.from_bytes(si[10:18], "big") & ((1 << 36) - 1)) == total,
                                "min_framesize_field": int.from_bytes(si[4:7], "big"), "max_framesize_field": int.from_bytes(si[7:10], "big")}
    ok = line["prefix_check"]["identical"] and line["streaminfo_check"]["md5_ok"] and line["streaminfo_check"]["total_samples_ok"] and line["crc16_bad_frame"] == -1
    line["all_checks_passed"] = bool(ok)
    if not args.keep and os.path.exists(flac):
        os.unlink(flac)
    with open(os.path.join(outdir, "corpus.json"), "w") as f:
        json.dump(line, f, indent=1)
    print(json.dumps(line))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
