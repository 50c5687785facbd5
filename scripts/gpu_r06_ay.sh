#!/bin/bash
# autoc3_kernel by window-job SETS (default) against by JOBS (FLACGPU_AUTOC3_SETS=0) over batch sizes: the sets' grid is 3 equal wavefronts per group
# (16384 frames: 3072 on 2048 slots = two rounds for 1.5 of work), the jobs' 6 of three lengths, longest first
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for frames in 8192 12288 16384 20480 24576 32768 49152 65536 262144; do
  steps=$((400000 / frames + 4))
  for r in 1 2; do
    for v in "X=1" "FLACGPU_AUTOC3_SETS=0"; do
      echo -n "$frames $v: "; env $v python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras --no-api --no-decode --no-clock --frames $frames 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'], d['kernel_ms'], d['verified']['ok'])"
    done
  done
done 2>&1 | tee $OUT/autoc3_sets_ab.txt
