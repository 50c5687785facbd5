#!/bin/bash
# round 2, visit D: verify kernel after the bit-reader rework, grouped autocorrelation v2 (one-wave workgroups, XCD-aware),
# scan v2, and A/B of the evaluation kernel's shape / candidate staging.
set -u
TAG=${1:-r02_d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 20"
run() { env "$@" timeout 300 $B --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['kernel_ms'])"; }
timeout 300 $B 2>$OUT/bench.err > $OUT/bench.json; python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['kernel_ms'],d.get('device_verify'),d['verified']['ok'])"
run X=1
run FLACGPU_AUTOC2_UNGROUPED=1
run FLACGPU_EVAL_CANDS_GLOBAL=1
run FLACGPU_EVAL_CPW=1 FLACGPU_EVAL_WAVES=4
run FLACGPU_EVAL_CPW=1 FLACGPU_EVAL_WAVES=8
run FLACGPU_EVAL_CPW=2 FLACGPU_EVAL_WAVES=4
run FLACGPU_EVAL_CPW=1 FLACGPU_EVAL_WAVES=4 FLACGPU_EVAL_CANDS_GLOBAL=1
run FLACGPU_FUSED_COMPACT=1
run X=2
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > /dev/null 2> $OUT/pmc.err
  DB=$(ls $OUT/pmc/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
  rm -rf $OUT/pmc
done
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc_counters.txt | grep -v "copyBuffer\|elementwise\|fillBuffer"
