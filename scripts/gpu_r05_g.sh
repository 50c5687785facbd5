#!/bin/bash
mkdir -p gpurun_out/r05_g
export TMPDIR=/tmp
timeout 120 build/ubench_mfma_fir > gpurun_out/r05_g/mfma_fir_ubench.txt 2>&1
cat gpurun_out/r05_g/mfma_fir_ubench.txt
(timeout 900 python -m pytest tests/test_headline_selection_gpu.py -m gpu -q --durations=25 -k "independent or lane_per_subframe_autocorrelation_of" 2>&1 | tail -45) > gpurun_out/r05_g/pytest_new.log 2>&1
tail -40 gpurun_out/r05_g/pytest_new.log
