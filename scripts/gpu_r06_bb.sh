#!/bin/bash
# evalw_kernel takes planes of 16-bit pairs: tests, parity files, the wasted-bits rate table
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 900 python -m pytest tests/test_pairs_in_wide_streams_gpu.py tests/test_gpu_parity.py tests/test_adversarial_gpu.py tests/test_block_sizes_fast_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($SECONDS s)"; tail -6 $OUT/pytest.log
timeout 300 python scripts/wasted_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/wasted_rate.txt
