#!/bin/bash
mkdir -p gpurun_out/r05_i
export TMPDIR=/tmp
timeout 300 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_i/order_rate.txt
cd /tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_i/prof -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-clock > gpurun_out/r05_i/prof_bench.json 2> gpurun_out/r05_i/prof.err
DB=$(ls gpurun_out/r05_i/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > gpurun_out/r05_i/kernel_stats_level8.txt && head -14 gpurun_out/r05_i/kernel_stats_level8.txt
rm -rf gpurun_out/r05_i/prof
