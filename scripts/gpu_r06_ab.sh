#!/bin/bash
# ff_kernel<., ., WIDE> (17..24-bit stereo at -0..-2 in one kernel): its tests, the ff / wide-decide / parity files, the presets x
# formats matrix with and without it, and a same-box A/B of the 16-bit -0 step against the previous build (build/alt_lib)
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 900 python -m pytest tests/test_wide_ff_gpu.py tests/test_wide_decide_gpu.py -x -q -m gpu > $OUT/pytest_wide.log 2>&1; echo "pytest wide rc=$? ($SECONDS s)"; tail -15 $OUT/pytest_wide.log
SECONDS=0
python scripts/matrix_rate.py 8192 2>&1 | grep -v amdgpu.ids | tee $OUT/matrix_rate.txt
FLACGPU_NO_WIDE_FF=1 python scripts/matrix_rate.py 8192 2>&1 | grep -v amdgpu.ids | grep "24-bit stereo" | tee $OUT/matrix_rate_no_wide_ff.txt
echo "matrix ($SECONDS s)"
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_adversarial_gpu.py tests/test_headline_selection_gpu.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$? ($SECONDS s)"; tail -4 $OUT/pytest_parity.log
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 3 --no-api --no-decode --no-clock --level 0 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level0.txt
