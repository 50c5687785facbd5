#!/bin/bash
# round 2, visit N: pack2 touching its subframes' sample lines ahead of the loads -- A/B on one box
set -u
TAG=${1:-r02_n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 180 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do for pf in 0 1; do for L in 8 5; do
  FLACGPU_PACK_PREFETCH=$pf timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --level $L > $OUT/b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('$OUT/b.json'))
print('prefetch $pf level $L: value %.0f ms %.4f pack %.4f' % (d['value'], d['ms_per_step'], d['kernel_ms']['pack']))
PY
done; done; done | tee $OUT/prefetch_ab.txt
