#!/usr/bin/env python3
"""Development aid: per kernel of a HIP source file, how many vector loads the compiler follows AT ONCE with s_waitcnt vmcnt(0) --
the signature of a load under a condition (or of a copy loop with its own bound), which serialises what could be in flight together.
usage: scripts/find_serial_loads.py flac_amd/csrc/flacgpu_verify.hip [kernel-name-substring ...]"""
import re
import subprocess
import sys

src, pats = sys.argv[1], sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                      "-fconstexpr-steps=50000000", "-Iinclude", "-Iflac_amd/csrc", "-S", "--cuda-device-only", src, "-o", "-"],
                     capture_output=True, text=True).stdout.split("\n")
start = None
for i, line in enumerate(asm):
    if line.startswith("_ZN7flacgpu") and ": " in line and "; @" in line:
        start, name = i, line.split(":")[0]
    elif start is not None and "s_endpgm" in line:
        body = asm[start:i]
        start = None
        if pats and not any(p in name for p in pats):
            continue
        loads = [k for k, l in enumerate(body) if re.search(r"\b(global|flat|buffer)_load", l)]
        serial = 0
        for k in loads:
            for j in range(k + 1, min(k + 12, len(body))):
                if re.search(r"(global|flat|buffer)_load", body[j]):
                    break
                if "s_waitcnt vmcnt(0)" in body[j]:
                    serial += 1
                    break
        print("%-70s loads %4d   followed at once by vmcnt(0): %3d" % (name[:70], len(loads), serial))
