#!/bin/bash
# experiment: eval with the planes staged into LDS by LDS-DMA (FLACGPU_EVAL_STAGE=1): parity subset, then on/off at -8 and -5
set -u
OUT=gpurun_out/${1:-r02_stage}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLACGPU_EVAL_STAGE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 240 -k "oracle_per_frame or golden or short_last or config3" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for L in 8 5; do for v in 0 1 0 1; do
  if [ $v = 1 ]; then export FLACGPU_EVAL_STAGE=1; else unset FLACGPU_EVAL_STAGE; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify --level $L 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('stage $v level $L', round(d['value']/1000,2), d['ms_per_step'], 'eval', d['kernel_ms']['eval'])"
done; done | tee $OUT/stage_ab.txt
