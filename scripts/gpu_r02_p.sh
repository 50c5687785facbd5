#!/bin/bash
# round 2, visit P: the committed state -- full GPU suite, bench line, kernel trace, PMC passes (traffic + VALU), smoke
set -u
TAG=${1:-r02_p}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 240 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | head -c 1200; echo; tail -2 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt && head -16 $OUT/kernel_stats.txt
rm -rf $OUT/prof
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > /dev/null 2> $OUT/pmc$i.err
  DB=$(ls $OUT/pmc$i/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
  rm -rf $OUT/pmc$i
done
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc_counters.txt | grep -v "copyBuffer\|elementwise\|fillBuffer"
