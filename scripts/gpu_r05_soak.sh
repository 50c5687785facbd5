#!/bin/bash
# round 5: a longer seeded sweep than the suite's 200 seeds -- 1000 seeds (~10 000 configurations) under the default kernel selection,
# and 600 with the streaming autocorrelation kernels forced (autoc3_kernel incl. <IND>, autoc4_kernel at test sizes)
mkdir -p gpurun_out/r05_soak
export TMPDIR=/tmp
(time FLACGPU_TEST_SEEDS=1000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" -x 2>&1 | tail -6) > gpurun_out/r05_soak/default.log 2>&1
(time FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 FLACGPU_TEST_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" -x 2>&1 | tail -6) > gpurun_out/r05_soak/forced.log 2>&1
cat gpurun_out/r05_soak/default.log gpurun_out/r05_soak/forced.log
