#!/bin/bash
mkdir -p gpurun_out/r05_t
export TMPDIR=/tmp
(time FLACGPU_ADV_SEEDS=${1:-120} timeout 400 python -m pytest tests/test_adversarial_gpu.py -m gpu -q 2>&1 | tail -25) > gpurun_out/r05_t/adversarial.log 2>&1
cat gpurun_out/r05_t/adversarial.log
