#!/bin/bash
# On the GPU box: bench the working-tree engine and each build/<name>/libflacgpu.so given.  usage: scripts/gpu_abn.sh name...
cp flac_amd/lib/libflacgpu.so /tmp/main.so
for v in main "$@"; do
  if [ $v = main ]; then cp /tmp/main.so flac_amd/lib/libflacgpu.so; else cp build/$v/libflacgpu.so flac_amd/lib/libflacgpu.so; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])"
done
cp /tmp/main.so flac_amd/lib/libflacgpu.so
