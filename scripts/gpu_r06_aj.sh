#!/bin/bash
# occupancy experiments by switch: evalg_kernel with two wavefronts per channel image (5 instead of 4 per SIMD), same box, alternating
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-api --no-decode --no-clock --frames 65536"
for r in 1 2; do
  for v in "" "FLACGPU_EVAL_WPC=2"; do
    echo "== -8 $v"; env $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['kernel_ms'], d['verified']['ok'])"
  done
done 2>&1 | tee $OUT/wpc_ab.txt
