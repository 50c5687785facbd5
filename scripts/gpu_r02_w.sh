#!/bin/bash
# round 2, visit W: crc_check / model with unconditional loads; where the eval workgroup's prologue goes
set -u
OUT=gpurun_out/${1:-r02_w}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_gpu_parity.py -x -q -m gpu --timeout 240 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $OUT/stamps.txt > /dev/null
grep "eval stamps\|eval prologue" $OUT/stamps.txt | tail -2
for L in 8 5; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --level $L > $OUT/b$L.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/b$L.json')); print('level $L', d['value'], d['ms_per_step'], d['kernel_ms'], d['device_verify']['ms_per_batch'], d['device_verify']['ms_per_batch_lane_per_frame'])"; done
