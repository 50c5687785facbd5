#!/bin/bash
# round 2, visit K: parked engine (a process's next stream reuses the previous one's engine and page-locked slots), MD5 chain
# that carries on while the engine comes up; API/CLI tests and rates
set -u
TAG=${1:-r02_k}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stream_encoder_api.py tests/test_flac_cli.py tests/test_dropin.py tests/test_verify_gpu.py -x -q -m gpu --timeout 180 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
gcc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,$PWD/flac_amd/lib && {
  echo "== md5 off"; FLACGPU_HOST_TIMING=1 ./build/api_rate 32768 8 0 2>&1 | grep -E "timing|level"
  echo "== md5 off, 1 stage thread"; FLACGPU_STAGE_THREADS=1 FLACGPU_HOST_TIMING=1 ./build/api_rate 32768 8 0 2>&1 | grep -E "timing|level"
  echo "== md5 off, no parking"; FLACGPU_ENGINE_CACHE=0 FLACGPU_HOST_TIMING=1 ./build/api_rate 32768 8 0 2>&1 | grep -E "timing|level"
  for b in 8 32; do echo "== batch budget $b MiB"; FLACGPU_HOST_TIMING=1 FLACGPU_BATCH_BYTES=$((b*1048576)) ./build/api_rate 32768 8 0 2>&1 | grep -E "timing|level"; done
  echo "== md5 on"; FLACGPU_HOST_TIMING=1 ./build/api_rate 32768 8 1 2>&1 | grep -E "timing|level"
  echo "== level 5"; ./build/api_rate 32768 5 0 2>&1 | grep -E "level"
} | tee $OUT/api_rate.txt
timeout 600 python scripts/cli_rate.py 30 2>&1 | tee $OUT/cli_rate.txt
