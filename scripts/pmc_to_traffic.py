#!/usr/bin/env python3
"""profiles/<tag>_pmc_counters.txt -> profiles/pmc_traffic.json: HBM-side bytes per frame and kernel.
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts 128-byte read
requests as 64 bytes; calibrated here on prep2_kernel, which must read the whole 8 B/sample PCM batch exactly once).
Also per kernel: SQ_INSTS_VALU per launch and per inter-channel sample (wavefront instructions: the unit of the VALU issue roofline,
1024 SIMDs x clock / 4 cycles), SQ_INSTS_SALU, and the share of cycles the kernel issues VALU work.
usage: pmc_to_traffic.py <pmc_counters.txt> <frames per launch> <tag> [blocksize=4096] [out=profiles/pmc_traffic.json]"""
import json, re, sys
src, frames, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
blocksize = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
outpath = sys.argv[5] if len(sys.argv) > 5 else "profiles/pmc_traffic.json"
vals, calls = {}, {}
for line in open(src):
    m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES)\s+n=(\d+)\s+avg=([0-9.e+]+)", line)
    if not m:
        continue
    name = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("flacgpu::", "")
    name = re.sub(r"<.*", "", name)
    # several instantiations of a kernel in one run (pack2_kernel<.., HINTS> of the verify step next to the timed one): the one with the
    # most launches is the timed one; later passes of the same instantiation overwrite earlier ones
    n = int(m.group(3))
    if n >= calls.get((name, m.group(2)), 0):
        calls[(name, m.group(2))] = n
        vals.setdefault(name, {})[m.group(2)] = float(m.group(4))
out = {"source": src, "tag": tag, "frames_per_launch": frames, "blocksize": blocksize,
       "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)", "kernels": {}}
for k, v in sorted(vals.items()):
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    e = {"fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"], "hbm_bytes_per_launch": int(hbm), "hbm_bytes_per_frame": hbm / frames}
    if "SQ_INSTS_VALU" in v:
        e["valu_wave_insts_per_launch"] = v["SQ_INSTS_VALU"]
        e["valu_wave_insts_per_sample"] = round(v["SQ_INSTS_VALU"] / (frames * blocksize), 4)
    if "SQ_INSTS_SALU" in v:
        e["salu_insts_per_sample"] = round(v["SQ_INSTS_SALU"] / (frames * blocksize), 4)
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["valu_busy_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
    out["kernels"][k] = e
out["hbm_bytes_per_frame_all_kernels"] = round(sum(k["hbm_bytes_per_frame"] for k in out["kernels"].values()), 1)
json.dump(out, open(outpath, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
