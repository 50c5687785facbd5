#!/bin/bash
# eval workgroup timeline at -5 and -8
set -u
OUT=gpurun_out/${1:-r02_e5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 5 8; do FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --level $L 2> $OUT/s$L.txt > /dev/null; echo "level $L"; grep "eval stamps\|eval variant\|pack2 stamps" $OUT/s$L.txt | tail -3; done
