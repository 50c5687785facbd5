#!/bin/bash
# soak of the one-kernel frame (ff_kernel, 16-bit and wide) on adversarial signals without the verify pass
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_FF_SEEDS=${2:-1500} timeout 1500 python -m pytest tests/test_wide_ff_gpu.py -m gpu -q -x -k adversarial 2>&1 | tail -12) > $OUT/ff_adversarial.log 2>&1
cat $OUT/ff_adversarial.log
