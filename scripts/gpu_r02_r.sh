#!/bin/bash
# round 2, visit R: kernel trace of the verify passes
set -u
TAG=${1:-r02_r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt && head -24 $OUT/kernel_stats.txt
rm -rf $OUT/prof
