#!/bin/bash
# round 2, visit Q: the hinted verify pass (a thread per 16-sample run, from the pack kernel's hints)
set -u
TAG=${1:-r02_q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py -x -q -m gpu --timeout 240 > $OUT/pytest_verify.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_verify.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_stream_encoder_api.py tests/test_flac_cli.py -x -q -m gpu --timeout 240 > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_rest.log
timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['kernel_ms'])
print('verify',d.get('device_verify'))
PY
tail -3 $OUT/bench.err
