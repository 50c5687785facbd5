#!/bin/bash
# round 2: per-kernel times of every preset (16384 frames per step)
set -u
OUT=gpurun_out/${1:-r02_levels}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 0 1 2 3 4 5 6 7 8; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify --level $L > $OUT/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/b.json')); print('level $L', round(d['value']/1000,2), 'G', d['ms_per_step'], d['kernel_ms'], d['config']['blocksize'])"; done | tee $OUT/levels.txt
