#!/bin/bash
# pack2_kernel<., false, 128 / 256, 18>: blocks of two / four 1152-sample passes on two / four wavefronts -- tests, then pack times with and without (FLACGPU_NO_RUN18W=1)
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_block_sizes_fast_gpu.py tests/test_prep3_block_sizes_gpu.py tests/test_headline_selection_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($SECONDS s)"; tail -4 $OUT/pytest.log
for r in 1 2; do
for v in X=1 FLACGPU_NO_RUN18W=1; do
env $v timeout 600 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | grep -- "-b 2304\|-b 4608\|-b 1152" | sed "s/^/$v /" | tee -a $OUT/order_rate.txt
env $v timeout 600 python scripts/chan_block_rate.py 2>&1 | grep -v amdgpu.ids | grep -- "-b 2304\|-b 4608" | sed "s/^/$v /" | tee -a $OUT/chan_block_rate.txt
done
done
