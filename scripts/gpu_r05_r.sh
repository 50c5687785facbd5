#!/bin/bash
mkdir -p gpurun_out/r05_r
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "bench or side_figures or callers_arrays or multi_rank or gpus_flag or self_launched or forced" 2>&1 | tail -8 | tee gpurun_out/r05_r/pytest.log
