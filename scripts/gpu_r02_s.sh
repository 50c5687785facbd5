#!/bin/bash
# round 2, visit S: stamps of the hinted verify workgroups
set -u
OUT=gpurun_out/${1:-r02_s}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/stamps.txt
grep "hinted verify stamps" $OUT/stamps.txt | tail -2
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['device_verify'])"
