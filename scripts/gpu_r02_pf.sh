#!/bin/bash
# does the successor prefetch of the eval kernel reach its loads? stamps with and without it, -8 and -5
set -u
OUT=gpurun_out/${1:-r02_pf}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 8 5; do for pf in default 0 8 128; do
  if [ $pf = default ]; then unset FLACGPU_EVAL_PREFETCH; else export FLACGPU_EVAL_PREFETCH=$pf; fi
  FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --level $L 2> $OUT/s.txt > /dev/null
  echo "level $L prefetch $pf: $(grep 'eval stamps' $OUT/s.txt | tail -1 | cut -c60-)"
done; done | tee $OUT/pf_stamps.txt
