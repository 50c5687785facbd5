#!/bin/bash
# where the time between the kernels goes: dispatch timelines of -5, -0 and -8 steps
mkdir -p gpurun_out/r05_p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_p
for lv in 5 0 8; do
  timeout 300 rocprofv3 --kernel-trace -d $O/kt$lv -o kt -- python bench.py --level $lv --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-verify --no-clock > $O/bench_$lv.json 2> $O/bench_$lv.err
  DB=$(ls $O/kt$lv/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_timeline.py $DB 36 > $O/timeline_$lv.txt
  rm -rf $O/kt$lv
  tail -30 $O/timeline_$lv.txt
done
