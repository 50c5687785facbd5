#!/bin/bash
# round 2, visit V: the whole GPU suite with poisoned scratch memory (FLACGPU_POISON=1), then normally, verify first
set -u
OUT=gpurun_out/${1:-r02_v}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLACGPU_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_stream_encoder_api.py tests/test_verify_gpu.py -q -m gpu --timeout 300 > $OUT/pytest_poison.log 2>&1; echo "poison rc=$?"; tail -8 $OUT/pytest_poison.log
timeout 1500 python -m pytest tests/test_verify_gpu.py tests/test_gpu_parity.py -x -q -m gpu --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
