"""`flac -t` for a set of files on the device: python -m flac_amd.flactest [--json] file.flac ...

What the reference's tool does per file in test mode (src/flac/decode.c: DecoderSession_process :501 -- decode everything, count the
error callbacks, let FLAC__stream_decoder_finish compare the MD5 of the decoded samples with STREAMINFO's, src/libFLAC/
stream_decoder.c:670-676), batched: every file is copied to the device and decoded there (flacgpu_decode_stream_device: frames found
by sync code, a lane per frame), its samples narrowed to the byte format of the digest (flacgpu_pack_samples_device), and ALL files'
MD5 chains run at once, a lane per file (flacgpu_md5_many_device).  Verdict per file as the tool words it: ok, or the first error /
"MD5 signature mismatch"; exit status 1 when any file fails.  No CPU decode path: without the HIP library or a GPU this fails loudly.
Python is plumbing (files, torch tensors for device memory); the work is the C ABI's."""
import argparse
import json
import sys
import time

import numpy as np

from .engine import FlacGpuError, md5_many_device
from .stream_decoder import ERROR_NAMES, StreamDecoder, probe


def test_files(paths, device=0):
    """[(path, verdict dict)] in the order given; verdict: ok, errors (names), md5 ('ok' | 'mismatch' | 'unset'), samples, seconds"""
    import torch
    dev = torch.device("cuda", device)
    dec = StreamDecoder(device)
    out = []
    packed = []                     # per file: (index into out, device byte tensor)
    t0 = time.perf_counter()
    try:
        for path in paths:
            v = dict(ok=False, errors=[], md5="unset", samples=0, channels=0, bps=0, note="")
            out.append((path, v))
            try:
                data = open(path, "rb").read()
            except OSError as e:
                v["note"] = "cannot read: %s" % e
                continue
            try:
                si, first, total, md5 = probe(data)
            except FlacGpuError:
                v["note"] = "metadata runs past the end of the file"
                continue
            if not si.has_streaminfo:
                v["note"] = "no STREAMINFO (not a FLAC file?)"
                continue
            n = len(data)
            d_stream = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
            d_stream[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
            values = max(int(total) * int(si.channels), 1)
            for attempt in range(2):
                d_pcm = torch.empty(values, dtype=torch.int32, device=dev)
                rc, res, events = dec.decode_device(d_stream.data_ptr(), n, first, si, d_pcm.data_ptr(), values, 64)
                if rc == -4:                                             # FLACGPU_ERR_OUTPUT_TOO_SMALL: STREAMINFO's total was 0 or wrong
                    values = int(res.samples) * int(res.channels)
                    continue
                break
            if rc != 0:
                v["note"] = "decode failed (%d)" % rc
                continue
            v["samples"], v["channels"], v["bps"] = int(res.samples), int(res.channels), int(res.bits_per_sample)
            v["errors"] = [ERROR_NAMES.get(e[0], str(e[0])) for e in events]
            if res.nevents > len(events):
                v["errors"].append("... %d more" % (res.nevents - len(events)))
            if res.end_in_header:
                v["errors"].append("stream ends inside a frame header")
            if res.format_changes:
                v["errors"].append("%d frames in another format" % res.format_changes)
            if total and int(res.samples) != int(total):
                v["errors"].append("decoded %d samples, STREAMINFO says %d" % (int(res.samples), int(total)))
            if md5 != bytes(16):
                nv = int(res.samples) * int(res.channels)
                d_bytes = torch.empty(max(nv * ((v["bps"] + 7) // 8), 1) + 64, dtype=torch.uint8, device=dev)
                dec.pack_samples(d_pcm.data_ptr(), nv, v["bps"], d_bytes.data_ptr())
                packed.append((len(out) - 1, d_bytes, nv * ((v["bps"] + 7) // 8), md5))
            del d_pcm, d_stream
        torch.cuda.synchronize(dev)
        if packed:
            # one launch for all the digests: the ranges are addressed from the lowest tensor's base
            base = min(t.data_ptr() for _, t, _, _ in packed)
            base -= base % 4
            offsets = [t.data_ptr() - base for _, t, _, _ in packed]
            lengths = [nb for _, _, nb, _ in packed]
            digests = md5_many_device(base, offsets, lengths, device=device)
            for (i, _, _, want), got in zip(packed, digests):
                out[i][1]["md5"] = "ok" if got == want else "mismatch"
        for _, v in out:
            v["ok"] = not v["errors"] and not v["note"] and v["md5"] != "mismatch"
    finally:
        dec.close()
    return out, time.perf_counter() - t0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("files", nargs="+")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args(argv)
    res, secs = test_files(a.files, a.device)
    bad = 0
    if a.json:
        print(json.dumps(dict(seconds=round(secs, 4), files=[dict(path=p, **v) for p, v in res])))
    for p, v in res:
        if not v["ok"]:
            bad += 1
        if not a.json:
            why = v["note"] or (", ".join(v["errors"]) if v["errors"] else "MD5 signature mismatch" if v["md5"] == "mismatch" else "")
            print("%s: %s" % (p, "ok" + ("" if v["md5"] != "unset" else " (no MD5 in STREAMINFO)") if v["ok"] else "ERROR " + why))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
