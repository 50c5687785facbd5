#!/bin/bash
# prep3_kernel<., NW, CH>: the stereo mid/side prep kernel at 1024 / 2048 / 8192 / 1152 / 2304 / 4608-sample blocks -- its tests, the block-size and parity files, then the rates with and without it
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_prep3_block_sizes_gpu.py -x -q -m gpu --durations=5 > $OUT/pytest_prep3.log 2>&1; echo "pytest prep3 rc=$? ($SECONDS s)"; tail -15 $OUT/pytest_prep3.log
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_block_sizes_fast_gpu.py tests/test_gpu_parity.py tests/test_headline_selection_gpu.py tests/test_adversarial_gpu.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_parity.log
for r in 1 2; do
timeout 600 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | grep -- "-b\|l 12" | tee -a $OUT/order_rate.txt
FLACGPU_NO_PREP3N=1 timeout 600 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | grep -- "-b" | tee -a $OUT/order_rate_no_prep3n.txt
done
