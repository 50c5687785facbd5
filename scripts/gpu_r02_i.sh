#!/bin/bash
# round 2, visit I: prep3 without register spills (traffic), host API with helper threads for the narrowing copy and the MD5
# chain running ahead of the submissions; where the single-stream wall time goes (FLACGPU_HOST_TIMING)
set -u
TAG=${1:-r02_i}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_stream_encoder_api.py tests/test_dropin.py -x -q -m gpu --timeout 180 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
gcc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,$PWD/flac_amd/lib && {
  for t in 1 2 4 8; do echo "== stage threads $t, md5 off"; FLACGPU_HOST_TIMING=1 FLACGPU_STAGE_THREADS=$t ./build/api_rate 32768 8 0 2>&1 | tail -2; done
  for b in 16 32 128; do echo "== batch budget $b MiB, 4 threads, md5 off"; FLACGPU_HOST_TIMING=1 FLACGPU_BATCH_BYTES=$((b*1048576)) ./build/api_rate 32768 8 0 2>&1 | tail -2; done
  echo "== md5 on"; FLACGPU_HOST_TIMING=1 ./build/api_rate 32768 8 1 2>&1 | tail -2
  echo "== md5 on, 16 MiB"; FLACGPU_HOST_TIMING=1 FLACGPU_BATCH_BYTES=16777216 ./build/api_rate 32768 8 1 2>&1 | tail -2
} | tee $OUT/api_rate.txt
timeout 600 python scripts/cli_rate.py 30 2>&1 | tee $OUT/cli_rate.txt
FLACGPU_BATCH_BYTES=16777216 timeout 600 python scripts/cli_rate.py 30 2>&1 | tail -5 | tee $OUT/cli_rate_16m.txt
timeout 600 python bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['kernel_ms'])
print('verify',d.get('device_verify'),'ok',d['verified']['ok'])
for k in ('white_noise','level5'): print(k,d[k]['value'],d[k]['kernel_ms'],d[k]['verified_ok'])
print(d['cpu_baseline'])
PY
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > /dev/null 2> $OUT/pmc.err
  DB=$(ls $OUT/pmc/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
  rm -rf $OUT/pmc
done
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc_counters.txt | grep -v "copyBuffer\|elementwise\|fillBuffer"
