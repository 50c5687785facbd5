#!/bin/bash
# One GPU-box visit: GPU parity tests, the bench line, a rocprofv3 kernel-trace summary of the same bench
# command, and (optionally) the PMC counter passes.  Everything lands under gpurun_out/<tag>/ as text.
# usage: scripts/gpu_round.sh <tag> [tests] [bench] [prof] [pmc]
set -u
TAG=${1:-round}; shift || true
WHAT="${*:-tests bench prof}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in $WHAT; do
  case $w in
    tests)
      timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log ;;
    bench)
      timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json ;;
    prof)
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
      echo "prof rc=$?"
      DB=$(ls $OUT/prof/*.db 2>/dev/null | head -1)
      [ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
      cat $OUT/prof_bench.json
      rm -f $OUT/prof/*.db ;;
    pmc)
      i=0
      for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
                 "SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT" \
                 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1))
        timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames 4096 > $OUT/pmc$i.json 2> $OUT/pmc$i.err
        echo "pmc pass $i rc=$? : $SET"
        DB=$(ls $OUT/pmc$i/*.db 2>/dev/null | head -1)
        [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
        rm -rf $OUT/pmc$i
      done
      cat $OUT/pmc_counters.txt ;;
  esac
done
