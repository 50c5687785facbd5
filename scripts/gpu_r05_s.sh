#!/bin/bash
# the seeded sweep with independent channels going a wavefront per window-job set at every size (FLACGPU_AUTOC3_IND_SETS=1) and the
# streaming autocorrelation kernels forced
mkdir -p gpurun_out/r05_s
export TMPDIR=/tmp
(time FLACGPU_AUTOC3_IND_SETS=1 FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_TEST_SEEDS=400 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" -x 2>&1 | tail -6) > gpurun_out/r05_s/forced_ind_sets.log 2>&1
cat gpurun_out/r05_s/forced_ind_sets.log
