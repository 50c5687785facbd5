#!/bin/bash
# round 2, visit X: A/B on one box of the model kernel's load pattern (old library from git stash vs new) -- here simply: run
# bench -8 / -5 three times each and report the model and pack times
set -u
OUT=gpurun_out/${1:-r02_x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in 8 5; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --level $L > $OUT/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/b.json')); print('level $L', d['value'], d['ms_per_step'], d['kernel_ms'])"; done; done
