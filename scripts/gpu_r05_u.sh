#!/bin/bash
mkdir -p gpurun_out/r05_u
export TMPDIR=/tmp
(time FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_AUTOC3_IND_SETS=1 FLACGPU_ADV_SEEDS=${1:-300} timeout 100 python -m pytest tests/test_adversarial_gpu.py -m gpu -q -x 2>&1 | tail -25) > gpurun_out/r05_u/adversarial_forced.log 2>&1
cat gpurun_out/r05_u/adversarial_forced.log
