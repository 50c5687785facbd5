#!/bin/bash
# round 5, visit b: the GPU suite with autoc3_kernel really forced; evalg A/B (clz exponent, host divisor table); chan_rate baseline
mkdir -p gpurun_out/r05_b
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r05_b/pytest.log 2>&1
timeout 600 python scripts/ab_engine.py flac_amd/lib/libflacgpu_prev.so flac_amd/lib/libflacgpu.so 3 > gpurun_out/r05_b/ab_evalg.txt 2>&1
timeout 300 python scripts/ab_engine.py flac_amd/lib/libflacgpu_prev.so flac_amd/lib/libflacgpu.so 2 --hires > gpurun_out/r05_b/ab_evalw_hires.txt 2>&1
timeout 300 python scripts/chan_rate.py 16384 > gpurun_out/r05_b/chan_rate_before.txt 2>&1
tail -5 gpurun_out/r05_b/pytest.log
