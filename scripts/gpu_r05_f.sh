#!/bin/bash
# round 5, visit f (d again, prep4 LDS attribute fixed): prep4_kernel + autoc3_kernel<IND> with uniform fetch paths; the int8 MFMA FIR microbenchmark; A/B of the headline against the tree of visit b
mkdir -p gpurun_out/r05_f
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r05_f/pytest.log 2>&1
FLACGPU_NO_FAST1=1 timeout 300 python scripts/chan_rate.py 16384 > gpurun_out/r05_f/chan_rate_nofast1.txt 2>&1
timeout 300 python scripts/chan_rate.py 16384 > gpurun_out/r05_f/chan_rate.txt 2>&1
timeout 300 python scripts/chan_rate.py 4096 > gpurun_out/r05_f/chan_rate_4096.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_f/bench.json 2> gpurun_out/r05_f/bench.err
tail -5 gpurun_out/r05_f/pytest.log; cat gpurun_out/r05_f/chan_rate.txt gpurun_out/r05_f/mfma_fir_ubench.txt
