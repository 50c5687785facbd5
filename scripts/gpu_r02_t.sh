#!/bin/bash
# round 2, visit T: hinted verify with and without the input prefetch for the successor workgroup
set -u
OUT=gpurun_out/${1:-r02_t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for pf in 0 1024 2048 default; do
  if [ $pf = default ]; then unset FLACGPU_VERIFY_PREFETCH; else export FLACGPU_VERIFY_PREFETCH=$pf; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/b.json'))['device_verify']; print('prefetch $pf:', d['ms_per_batch'], d['frames_verified_a_thread_per_run'], d['status'])"
done; done | tee $OUT/verify_prefetch_ab.txt
unset FLACGPU_VERIFY_PREFETCH
FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "hinted verify stamps" | tail -1
