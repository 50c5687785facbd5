#!/bin/bash
# evalg_kernel<32>: predictors of 13..32 taps on the wavefront-per-channel evaluation -- its tests, the parity file, then the order rates
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_long_predictors_gpu.py -x -q -m gpu > $OUT/pytest_long.log 2>&1; echo "pytest long rc=$? ($SECONDS s)"; tail -15 $OUT/pytest_long.log
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_adversarial_gpu.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_parity.log
timeout 600 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/order_rate.txt
FLACGPU_NO_EVALG32=1 timeout 600 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | grep -- "-l" | tee $OUT/order_rate_no_evalg32.txt
