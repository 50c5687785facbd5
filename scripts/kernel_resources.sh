#!/bin/bash
# per-kernel register / LDS / scratch usage as the compiler reports it (development aid)
# usage: scripts/kernel_resources.sh flac_amd/csrc/flacgpu_analyze.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fconstexpr-steps=50000000 -Iinclude -Iflac_amd/csrc -c "$1" ${@:2} -o /tmp/kr.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = None
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): *(\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = "%-34s" % re.sub(r"^_ZN7flacgpu\d+", "", v)[:34]
    else: cur += " %s=%s" % (k.split(" [")[0].replace(" ", ""), v)
if cur: print(cur)
'
