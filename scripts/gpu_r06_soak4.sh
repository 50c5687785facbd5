#!/bin/bash
# the final tree (evalw_kernel on planes of pairs): the seeded sweep and the adversarial signals
mkdir -p gpurun_out/r06_soak4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(FLACGPU_TEST_SEEDS=${1:-800} timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -2) > gpurun_out/r06_soak4/sweep.log 2>&1
(FLACGPU_ADV_SEEDS=${2:-500} timeout 200 python -m pytest tests/test_adversarial_gpu.py -m gpu -q 2>&1 | tail -2) > gpurun_out/r06_soak4/adversarial.log 2>&1
cat gpurun_out/r06_soak4/sweep.log gpurun_out/r06_soak4/adversarial.log
