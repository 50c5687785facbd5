#!/bin/bash
# On the GPU box: alternate bench runs between the working-tree engine (A) and build/alt_lib (B); the order flips every round
# (the first run of a pair is consistently ~1.5% slower).  usage: scripts/gpu_ab.sh [rounds] [bench args]
R=${1:-3}; shift || true
cp flac_amd/lib/libflacgpu.so /tmp/A.so; cp build/alt_lib/libflacgpu.so /tmp/B.so
for i in $(seq $R); do
  if [ $((i % 2)) = 1 ]; then order="A B"; else order="B A"; fi
  for v in $order; do
    cp /tmp/$v.so flac_amd/lib/libflacgpu.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])"
  done
done
cp /tmp/A.so flac_amd/lib/libflacgpu.so
