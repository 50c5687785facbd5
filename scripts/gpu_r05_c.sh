#!/bin/bash
# round 5, visit c: autoc3_kernel<IND> + prep2 staging: the GPU suite, the channel-layout rates before (FLACGPU_NO_FAST1=1) and after, the bench line with the clock probe
mkdir -p gpurun_out/r05_c
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r05_c/pytest.log 2>&1
FLACGPU_NO_FAST1=1 timeout 300 python scripts/chan_rate.py 16384 > gpurun_out/r05_c/chan_rate_nofast1.txt 2>&1
timeout 300 python scripts/chan_rate.py 16384 > gpurun_out/r05_c/chan_rate.txt 2>&1
timeout 300 python scripts/chan_rate.py 4096 > gpurun_out/r05_c/chan_rate_4096.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_c/bench.json 2> gpurun_out/r05_c/bench.err
tail -5 gpurun_out/r05_c/pytest.log; cat gpurun_out/r05_c/chan_rate.txt
