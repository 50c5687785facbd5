#!/bin/bash
# the reference's test/test_flac.sh on the drop-in, to its end
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_SHELL_SUITE=full timeout 2400 python -m pytest tests/test_shell_suites_gpu.py -m gpu -q -x -s -k test_flac_sh 2>&1 | tail -40) > $OUT/shell_test_flac_full.log 2>&1
cat $OUT/shell_test_flac_full.log
