#!/bin/bash
# One GPU-box visit for a change under test: a -k selection of the GPU tests under a short timeout (a kernel that spins must not
# hold the box), then short bench lines.  usage: scripts/gpu_quick.sh <tag> "<pytest -k expr>" [bench args per line, ";"-separated]
TAG=$1; EXPR=$2; BENCHES=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 90 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" || { echo "GPU health check failed"; exit 3; }
if [ -n "$EXPR" ]; then
  FLACGPU_POISON=1 timeout ${QUICK_TEST_TIMEOUT:-420} python -m pytest tests -x -q -m gpu -k "$EXPR" > $OUT/pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -15 $OUT/pytest.log
  [ $rc -ne 0 ] && exit 4
fi
IFS=';' read -ra BL <<< "$BENCHES"
i=0
for b in "${BL[@]}"; do
  i=$((i+1))
  timeout 300 python bench.py --no-cpu-baseline --no-extras $b > $OUT/bench_$i.json 2> $OUT/bench_$i.err; echo "bench [$b] rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$i.json")); print(d["value"], d["ms_per_step"], d["kernel_ms"], d.get("verified"))
except Exception as e:
    print("no line", e); print(open("$OUT/bench_$i.err").read()[-1500:])
PY
done
