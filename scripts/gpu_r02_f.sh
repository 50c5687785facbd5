#!/bin/bash
# round 2, visit F: the persistent / pipelined evaluation kernel: parity and A/B
set -u
TAG=${1:-r02_f}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_stream_encoder_api.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 20"
run() { env "$@" timeout 300 $B --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['kernel_ms'])"; }
timeout 300 $B 2>$OUT/bench.err > $OUT/bench.json; python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['kernel_ms'],d['verified']['ok'])"
run X=1
run FLACGPU_NO_EVAL3=1
run FLACGPU_EVAL3_PER_CU=1
run FLACGPU_EVAL3_PER_CU=2
run FLACGPU_EVAL3_PER_CU=3
run FLACGPU_EVAL3_PER_CU=4
env X=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --level 7 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('l7', d['value'], d['kernel_ms'])"
env FLACGPU_NO_EVAL3=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --level 7 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('l7 no3', d['value'], d['kernel_ms'])"
gcc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,$PWD/flac_amd/lib && ./build/api_rate 32768 8 0 2>&1 | tail -1
