#!/bin/bash
# autoc3_kernel<IND> / autoc4_kernel: the batch's last group slid back to be whole -- the IND test files, then the channel x block-size rates
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_headline_selection_gpu.py tests/test_zz_channel_counts_gpu.py tests/test_long_predictors_gpu.py tests/test_prep3_block_sizes_gpu.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($SECONDS s)"; tail -4 $OUT/pytest.log
for r in 1 2; do
timeout 600 python scripts/chan_block_rate.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/chan_block_rate.txt
done
timeout 600 python scripts/chan_rate.py 16384 2>&1 | grep -v amdgpu.ids | tee -a $OUT/chan_rate.txt
timeout 600 python scripts/chan_rate.py 10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/chan_rate.txt
