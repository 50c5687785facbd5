#!/bin/bash
# evalg_kernel with two samples' chains interleaved per asm statement (EG_CHAIN2): same-box A/B against the previous build, parity
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 3 --no-api --no-decode --no-clock --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level8.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --level 5 --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level5.txt
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_long_predictors_gpu.py tests/test_block_sizes_fast_gpu.py tests/test_headline_selection_gpu.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_parity.log
