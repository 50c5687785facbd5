#!/bin/bash
# prep3_kernel<., DIRECT> (owner-layout loads, no LDS tile) against the LDS form (FLACGPU_PREP3_LDS=1): same box, alternating; parity
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-api --no-decode --no-clock --frames 65536"
for args in "" "--level 5" "--hires"; do
  for r in 1 2 3; do
    for v in "FLACGPU_PREP3_LDS=1" "X=1"; do
      echo -n "$args $v: "; env $v $B $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['kernel_ms'], d['verified']['ok'])"
    done
  done
done 2>&1 | tee $OUT/prep3_ab.txt
SECONDS=0
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_headline_selection_gpu.py tests/test_adversarial_gpu.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$? ($SECONDS s)"; tail -3 $OUT/pytest_parity.log
