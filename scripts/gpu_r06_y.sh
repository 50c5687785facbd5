#!/bin/bash
# autoc3_kernel with a tile of doubles: the tests that force it, then same-box A/B against the previous build (build/alt_lib)
TAG=$1; R=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_POISON=1 timeout 900 python -m pytest tests -x -q -m gpu -k "headline or autoc3 or golden or parity or adversarial or channel_counts or large_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so $R --no-api --no-decode --no-clock 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level8.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --hires --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_hires.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --level 5 --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level5.txt
