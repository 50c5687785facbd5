#!/bin/bash
# round 2, visit H: full GPU suite (per-test timeout) with the evaluation prefetch and the rewritten verify decode path; bench line
set -u
TAG=${1:-r02_h}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu --timeout 180 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['kernel_ms'])
print('verify',d.get('device_verify'),'ok',d['verified']['ok'])
for k in ('white_noise','level5'): print(k,d[k]['value'],d[k]['kernel_ms'],d[k]['verified_ok'])
PY
gcc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,$PWD/flac_amd/lib && { ./build/api_rate 32768 8 0 | tail -1; ./build/api_rate 32768 8 1 | tail -1; } | tee $OUT/api_rate.txt
