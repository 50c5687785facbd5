#!/bin/bash
# -5 at the bench's batch size: autoc3_kernel forced (one wavefront per SIMD) against the default selection (autoc2_kernel)
TAG=$1; R=${2:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --level 5 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --no-clock | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'])"; }
for i in $(seq $R); do
  run default X=1
  run autoc3 FLACGPU_AUTOC3=1
  run autoc3_sets FLACGPU_AUTOC3=1 FLACGPU_AUTOC3_SETS=1
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
