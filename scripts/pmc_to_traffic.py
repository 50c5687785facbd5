#!/usr/bin/env python3
"""profiles/<tag>_pmc_counters_<workload>.txt -> profiles/pmc_traffic.json: per workload and kernel, HBM-side bytes per frame,
wavefront instructions per inter-channel sample, and the share of cycles the kernel issues VALU work -- what bench.py prices its
lines with (roofline.traffic*, roofline_valu).

hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts 128-byte read requests as 64
bytes; calibrated on prep3_kernel, which must read the whole 8 B/sample PCM batch exactly once).

The file also records the git blob hash of every kernel source at the time of the passes: bench.py compares them with the tree it
runs from and marks a line "stale" when a kernel was edited after its counters were taken.

usage: pmc_to_traffic.py <tag> <workload>=<pmc_counters.txt>:<frames per launch>:<blocksize> [...]      (workloads: level8 level5 level0 white8 hires8)
"""
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = ("INT32", "INT64", "FMA_F64", "ADD_F64", "MUL_F64", "CVT", "FMA_F32", "ADD_F32")     # SQ_INSTS_VALU_<class>: the hardware's own instruction classes
COUNTERS = "FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAVES|" + "|".join("SQ_INSTS_VALU_" + c for c in CLASSES)


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    files = sorted(glob.glob(os.path.join(ROOT, "flac_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "flac_amd", "csrc", "*.h")))
    return {os.path.relpath(f, ROOT): blob_hash(f) for f in files}


def parse(src, frames, blocksize):
    vals, calls = {}, {}
    for line in open(src):
        m = re.match(r"(.*?)\s+(%s)\s+n=(\d+)\s+avg=([0-9.e+]+)" % COUNTERS, line)
        if not m:
            continue
        name = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("flacgpu::", "")
        name = re.sub(r"<.*", "", name)
        # several instantiations of a kernel in one run (pack2_kernel<.., HINTS> of the verify step next to the timed one): the one with
        # the most launches is the timed one; later passes of the same instantiation overwrite earlier ones
        n = int(m.group(3))
        if n >= calls.get((name, m.group(2)), 0):
            calls[(name, m.group(2))] = n
            vals.setdefault(name, {})[m.group(2)] = float(m.group(4))
    kernels = {}
    for k, v in sorted(vals.items()):
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or not k.endswith("_kernel"):
            continue
        hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        e = {"fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"], "hbm_bytes_per_frame": round(hbm / frames, 1)}
        if "SQ_INSTS_VALU" in v:
            e["valu_wave_insts_per_sample"] = round(v["SQ_INSTS_VALU"] / (frames * blocksize), 4)
        if "SQ_INSTS_VALU" in v and any("SQ_INSTS_VALU_" + c in v for c in CLASSES):
            # the class mix of the kernel's VALU instructions (round 6, VERDICT r05 #8: the issue roofline against per-class measured
            # cycles, scripts/ubench_cycles.hip); "other" = what the class counters do not claim (moves, lane exchange, bit operations ...)
            cl = {c.lower(): round(v.get("SQ_INSTS_VALU_" + c, 0.0) / (frames * blocksize), 4) for c in CLASSES}
            cl["other"] = round(max(0.0, v["SQ_INSTS_VALU"] - sum(v.get("SQ_INSTS_VALU_" + c, 0.0) for c in CLASSES)) / (frames * blocksize), 4)
            e["valu_class_per_sample"] = cl
        if "SQ_INSTS_SALU" in v:
            e["salu_insts_per_sample"] = round(v["SQ_INSTS_SALU"] / (frames * blocksize), 4)
        if "SQ_INSTS_LDS" in v:
            e["lds_insts_per_sample"] = round(v["SQ_INSTS_LDS"] / (frames * blocksize), 4)
        if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            e["valu_busy_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
        if "SQ_WAIT_INST_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            e["wait_inst_frac_of_wave_cycles"] = round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3)
        kernels[k] = e
    return {"source": os.path.relpath(src, ROOT) if os.path.isabs(src) else src, "frames_per_launch": frames, "blocksize": blocksize, "kernels": kernels,
            "hbm_bytes_per_frame_all_kernels": round(sum(k["hbm_bytes_per_frame"] for n, k in kernels.items() if not n.startswith(("verify", "crc_check"))), 1)}


def main():
    tag = sys.argv[1]
    out = {"tag": tag, "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)",
           "source_hashes": source_hashes(), "workloads": {}}
    for spec in sys.argv[2:]:
        name, rest = spec.split("=", 1)
        path, frames, bs = rest.split(":")
        out["workloads"][name] = parse(path, int(frames), int(bs))
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    for name, w in out["workloads"].items():
        print(name, w["hbm_bytes_per_frame_all_kernels"], {k: (v.get("valu_wave_insts_per_sample"), v["hbm_bytes_per_frame"]) for k, v in w["kernels"].items() if not k.startswith(("verify", "crc_check"))})


if __name__ == "__main__":
    main()
