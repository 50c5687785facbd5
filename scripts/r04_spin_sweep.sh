#!/bin/bash
# -0 / -2 / -8 / -5: two-kernel compaction | default (ff_kernel publishes + place kernel; pack2_kernel fused) | ff_kernel fused
for lv in 0 2 8 5; do
  for mode in nofuse default ff_fused; do
    unset FLACGPU_NO_FUSED_COMPACT FLACGPU_FF_FUSED
    [ $mode = nofuse ] && export FLACGPU_NO_FUSED_COMPACT=1
    [ $mode = ff_fused ] && export FLACGPU_FF_FUSED=1
    [ $mode = ff_fused ] && [ $lv -gt 2 ] && continue
    python bench.py --level $lv --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level $lv $mode', d['value'], d['ms_per_step'], d['kernel_ms'])"
  done
done
