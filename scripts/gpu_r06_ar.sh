#!/bin/bash
# sub-batches on their own streams (FLACGPU_SUBBATCHES=n: non-fused output) against the single stream with and without the fused output: same box, alternating
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for frames in 65536 262144; do
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-api --no-decode --no-clock --frames $frames"
for args in "" "--level 5" "--hires"; do
  for r in 1 2; do
    for v in "X=1" "FLACGPU_NO_FUSED_COMPACT=1" "FLACGPU_SUBBATCHES=2" "FLACGPU_SUBBATCHES=3" "FLACGPU_SUBBATCHES=4"; do
      echo -n "$frames $args $v: "; env $v $B $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d['kernel_ms'], d['verified']['ok'])"
    done
  done
done
done 2>&1 | tee $OUT/subbatch_ab.txt
