#!/bin/bash
# instruction-cache behaviour of the kernels (code of the evaluation kernel is large)
set -u
OUT=gpurun_out/${1:-r02_ic}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|SQ_IFETCH|INST_CACHE|SQC_" | head -40 > $OUT/counters.txt; head -30 $OUT/counters.txt
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > /dev/null 2> $OUT/pmc.err
  DB=$(ls $OUT/pmc/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/ic_counters.txt
  rm -rf $OUT/pmc
done
grep -E "ICACHE|IFETCH|WAIT_INST" $OUT/ic_counters.txt | grep -v "copyBuffer\|elementwise"
