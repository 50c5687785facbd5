#!/bin/bash
# prep4_kernel<., NW, CH>: independent channels at the block sizes of prep3_shape -- tests, the channel-count / selection / parity files, rates of mono and 5.1 with and without
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_prep3_block_sizes_gpu.py -x -q -m gpu --durations=5 > $OUT/pytest_prep34.log 2>&1; echo "pytest prep3/4 rc=$? ($SECONDS s)"; tail -12 $OUT/pytest_prep34.log
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_zz_channel_counts_gpu.py tests/test_block_sizes_fast_gpu.py tests/test_headline_selection_gpu.py tests/test_adversarial_gpu.py tests/test_wide_decide_gpu.py -x -q -m gpu > $OUT/pytest_more.log 2>&1; echo "pytest more rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_more.log
for r in 1 2; do
for v in X=1 FLACGPU_NO_PREP3N=1; do
env $v timeout 600 python scripts/chan_block_rate.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $OUT/chan_block_rate.txt
done
done
