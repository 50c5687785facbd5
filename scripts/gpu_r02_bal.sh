#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_bal}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 8 6 5; do FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --level $L 2> $OUT/s.txt > /dev/null; echo "level $L: $(grep 'candidate rounds' $OUT/s.txt | tail -1)"; done | tee $OUT/balance.txt
