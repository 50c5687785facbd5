#!/bin/bash
# the final tree: both reference shell suites to their end on the drop-in (their -b / -l matrices now reach prep3_kernel / prep4_kernel<., NW, CH>)
OUT=gpurun_out/r06_aw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_SHELL_SUITE=full timeout 2400 python -m pytest tests/test_shell_suites_gpu.py -m gpu -q -x -s 2>&1 | tail -8) > $OUT/shell_suites_full.log 2>&1; cat $OUT/shell_suites_full.log
