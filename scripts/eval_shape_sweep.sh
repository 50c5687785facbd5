#!/bin/bash
# On the GPU box: bench the evaluation kernel's workgroup shapes (channels per workgroup x wavefronts).
for shape in "0 0" "1 4" "2 4" "2 8" "4 4" "4 8" "1 8" "1 7" "2 7"; do
  set -- $shape
  FLACGPU_EVAL_CPW=$1 FLACGPU_EVAL_WAVES=$2 python bench.py --steps 8 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpw=$1 waves=$2', d['value'], d['kernel_ms']['eval'])"
done
