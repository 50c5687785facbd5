#!/bin/bash
# round 2, visit L: where the pack2 and eval workgroups spend their wall time (FLACGPU_DEBUG_TIMING stamps)
set -u
TAG=${1:-r02_l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 8 5; do
  FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --level $L > $OUT/bench_l$L.json 2> $OUT/stamps_l$L.txt
  echo "== level $L"; grep "flacgpu" $OUT/stamps_l$L.txt | tail -6
done
