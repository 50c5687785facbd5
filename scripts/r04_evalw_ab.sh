#!/bin/bash
for i in 1 2; do
  for wpc in 1 2; do
    FLACGPU_EVALW_WPC=$wpc python bench.py --hires --steps 10 --warmup 2 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hires wpc $wpc', d['value'], d['ms_per_step'], d['kernel_ms'], d['verified']['ok'])"
  done
done
