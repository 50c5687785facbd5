#!/bin/bash
# the final tree: the seeded sweep with autoc3_kernel forced under both grids (a wavefront per set / per job), damaged streams against the reference decoder
mkdir -p gpurun_out/r06_soak3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_AUTOC3_SETS=2 FLACGPU_TEST_SEEDS=${1:-500} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -3) > gpurun_out/r06_soak3/sweep_sets.log 2>&1
(time FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_AUTOC3_SETS=0 FLACGPU_TEST_SEEDS=${1:-500} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -3) > gpurun_out/r06_soak3/sweep_jobs.log 2>&1
(time FLACGPU_SD_SEEDS=${2:-1000} timeout 1500 python -m pytest tests/test_stream_decode_gpu.py -m gpu -q -k "damaged" 2>&1 | tail -3) > gpurun_out/r06_soak3/stream_decode.log 2>&1
cat gpurun_out/r06_soak3/sweep_sets.log gpurun_out/r06_soak3/sweep_jobs.log gpurun_out/r06_soak3/stream_decode.log
