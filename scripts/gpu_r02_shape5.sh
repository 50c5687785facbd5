#!/bin/bash
# eval workgroup shape sweep at -5 (and -3 / -0 for reference)
set -u
OUT=gpurun_out/${1:-r02_shape5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 5 3; do for shape in "0 0" "2 4" "4 8" "1 2" "1 4" "2 8" "4 4" "2 2" "1 1"; do set -- $shape
  FLACGPU_EVAL_CPW=$1 FLACGPU_EVAL_WAVES=$2 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify --level $L 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('level $L cpw $1 waves $2:', round(d['value']/1000,2), d['ms_per_step'], 'eval', d['kernel_ms']['eval'])"
done; done | tee $OUT/shape5.txt
