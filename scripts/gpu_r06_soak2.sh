#!/bin/bash
# soak of the tree with prep3_kernel / prep4_kernel<., NW, CH>, the slid last group of autoc3_kernel<IND> / autoc4_kernel and pack2_kernel<., ., 128 / 256, 18>:
# the seeded sweep (now drawing 1024 / 2048-sample blocks too), the same with autoc3_kernel forced, the adversarial signals
mkdir -p gpurun_out/r06_soak2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_TEST_SEEDS=${1:-800} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -6) > gpurun_out/r06_soak2/sweep.log 2>&1
(time FLACGPU_AUTOC3=1 FLACGPU_TEST_SEEDS=${1:-800} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -6) > gpurun_out/r06_soak2/sweep_autoc3.log 2>&1
(time FLACGPU_ADV_SEEDS=${2:-600} timeout 1500 python -m pytest tests/test_adversarial_gpu.py -m gpu -q 2>&1 | tail -8) > gpurun_out/r06_soak2/adversarial.log 2>&1
cat gpurun_out/r06_soak2/sweep.log gpurun_out/r06_soak2/sweep_autoc3.log gpurun_out/r06_soak2/adversarial.log
