#!/bin/bash
for lv in 0 2; do
  for mode in nofuse lag0 lag2048; do
    unset FLACGPU_NO_FUSED_COMPACT FLACGPU_FF_LAG
    case $mode in lag*) export FLACGPU_FF_LAG=${mode#lag};; esac
    python bench.py --level $lv --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level $lv $mode', d['value'], d['ms_per_step'], d['kernel_ms'])"
  done
done
