#!/bin/bash
# round 2, visit M: pack2 with the conflict-free CRC (44-byte spans, no byte-serial tail) and scalar decision fields
set -u
TAG=${1:-r02_m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_verify_gpu.py -x -q -m gpu --timeout 180 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for L in 8 5; do
  FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --level $L > /dev/null 2> $OUT/stamps_l$L.txt
  echo "== level $L"; grep "pack2 stamps" $OUT/stamps_l$L.txt | tail -1
done
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['kernel_ms'])
print('verified ok',d['verified']['ok'])
for k in ('white_noise','level5'): print(k,d[k]['value'],d[k]['kernel_ms'],d[k]['verified_ok'])
PY
