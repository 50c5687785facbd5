#!/bin/bash
# round 2: the fast presets -- prep2 without spills / with batched staging loads, autoc2 stereo-pair source, pack2 with 128 threads
# for 1152-sample blocks: parity, then A (working tree) / B (HEAD) per preset on one box
set -u
OUT=gpurun_out/${1:-r02_fast}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_verify_gpu.py -x -q -m gpu --timeout 240 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cp flac_amd/lib/libflacgpu.so /tmp/A.so; cp build/alt_lib/libflacgpu.so /tmp/B.so
for L in 0 1 2 3 4 5 8; do for v in A B A B; do
  cp /tmp/$v.so flac_amd/lib/libflacgpu.so
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify --level $L 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v level $L', round(d['value']/1000,2), d['ms_per_step'], d['kernel_ms'])"
done; done | tee $OUT/fast_presets_ab.txt
cp /tmp/A.so flac_amd/lib/libflacgpu.so
