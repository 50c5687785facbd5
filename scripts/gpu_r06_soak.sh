#!/bin/bash
# round 6 soaks: the seeded sweep and the adversarial signals under the default selection (new this round: evalg_kernel's 18- and
# 36-sample runs, pack2<run18> beyond the fixed presets, the wide deciding prep kernel), and the device stream decoder against the
# reference's decoder on damaged reference-written files
mkdir -p gpurun_out/r06_soak
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_TEST_SEEDS=${1:-800} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -6) > gpurun_out/r06_soak/sweep.log 2>&1
(time FLACGPU_ADV_SEEDS=${2:-600} timeout 1500 python -m pytest tests/test_adversarial_gpu.py -m gpu -q 2>&1 | tail -8) > gpurun_out/r06_soak/adversarial.log 2>&1
(time FLACGPU_SD_SEEDS=${3:-1500} timeout 1500 python -m pytest tests/test_stream_decode_gpu.py -m gpu -q -k "damaged" 2>&1 | tail -8) > gpurun_out/r06_soak/stream_decode.log 2>&1
cat gpurun_out/r06_soak/sweep.log gpurun_out/r06_soak/adversarial.log gpurun_out/r06_soak/stream_decode.log
