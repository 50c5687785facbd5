#!/bin/bash
# round 2: 24-bit / 96 kHz stereo at -8 and -5, per kernel
set -u
OUT=gpurun_out/${1:-r02_hires}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 8 5; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --hires --level $L > $OUT/b.json 2> $OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/b.json')); print('hires level $L', round(d['value']/1000,2), 'G', d['ms_per_step'], d['kernel_ms'], d.get('device_verify'))" || tail -3 $OUT/err.txt; done | tee $OUT/hires.txt
