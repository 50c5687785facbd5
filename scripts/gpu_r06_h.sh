#!/bin/bash
# same-box A/B by environment switches: the flat-window path of autoc3_kernel (FLACGPU_NO_FLAT=1: off) and the phase timing on
# every step / every fourth / never.  usage: scripts/gpu_r06_h.sh <tag> [rounds]
TAG=$1; R=${2:-4}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_POISON=1 timeout 600 python -m pytest tests -x -q -m gpu -k "headline or autoc3 or golden or parity" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify $BARGS | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'])"
}
for i in $(seq $R); do
  BARGS="--timing-every 1" run flat_t1 X=1
  BARGS="--timing-every 1" run noflat_t1 FLACGPU_NO_FLAT=1
  BARGS="--timing-every 4" run flat_t4 X=1
  BARGS="--timing-every 0" run flat_t0 X=1
done 2>&1 | tee $OUT/ab.txt
