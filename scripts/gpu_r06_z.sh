#!/bin/bash
# the running sum as v_sad_u32's third operand in prep2_chunk (prep2 / prep3 / prep4 / ff kernels): the whole GPU suite, then same-box
# A/B against the previous build (build/alt_lib)
TAG=$1; R=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
FLACGPU_POISON=1 timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($SECONDS s)"; tail -3 $OUT/pytest.log
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so $R --no-api --no-decode --no-clock 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level8.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --hires --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_hires.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --level 5 --frames 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level5.txt
python scripts/ab_engine.py build/alt_lib/libflacgpu.so flac_amd/lib/libflacgpu.so 2 --no-api --no-decode --no-clock --level 0 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_level0.txt
