#!/bin/bash
# round 2, visit G: L2 prefetch of the successor workgroup's inputs in the evaluation kernel: parity and A/B
set -u
TAG=${1:-r02_g}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 20"
run() { env "$@" timeout 200 $B --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['kernel_ms'])"; }
run X=1
run FLACGPU_EVAL_PREFETCH=0
run FLACGPU_EVAL_PREFETCH=32
run FLACGPU_EVAL_PREFETCH=64
run FLACGPU_EVAL_PREFETCH=96
run FLACGPU_EVAL_PREFETCH=128
run FLACGPU_EVAL_PREFETCH=256
run X=2
run5() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --level 5 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('l5 $*', d['value'], d['kernel_ms'])"; }
run5 X=1
run5 FLACGPU_EVAL_PREFETCH=0
run5 FLACGPU_EVAL_PREFETCH=64
run5 FLACGPU_EVAL_PREFETCH=256
