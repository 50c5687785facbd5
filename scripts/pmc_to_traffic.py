#!/usr/bin/env python3
"""profiles/<tag>_pmc_counters.txt -> profiles/pmc_traffic.json: HBM-side bytes per frame and kernel.
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE counts 128-byte read
requests as 64 bytes; calibrated here on prep2_kernel, which must read the whole 8 B/sample PCM batch exactly once).
usage: pmc_to_traffic.py <pmc_counters.txt> <frames per launch> <tag>"""
import json, re, sys
src, frames, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals = {}
for line in open(src):
    m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE|SQ_WAVE_CYCLES)\s+n=\d+\s+avg=([0-9.e+]+)", line)
    if not m:
        continue
    name = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("flacgpu::", "")
    name = re.sub(r"<.*", "", name)
    vals.setdefault(name, {})[m.group(2)] = float(m.group(3))     # later passes overwrite earlier ones
out = {"source": src, "tag": tag, "frames_per_launch": frames,
       "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)", "kernels": {}}
for k, v in sorted(vals.items()):
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    e = {"fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"], "hbm_bytes_per_launch": int(hbm), "hbm_bytes_per_frame": hbm / frames}
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["valu_busy_frac"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
    out["kernels"][k] = e
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
