#!/bin/bash
# One GPU-box visit, parameterised (replaces the per-visit scripts of earlier rounds).  Everything lands under
# gpurun_out/<tag>/ as text.  usage: scripts/gpu_visit.sh <tag> <step> [<step> ...]
#   tests            the GPU parity suite (FLACGPU_POISON=1)
#   tests:<expr>     pytest -k <expr>
#   bench[:args]     bench.py [args] -> bench<i>.json          (":" separates, "," stands for a blank inside args)
#   env:K=V          export K=V for the steps that follow (env:-K unsets)
#   prof[:args]      rocprofv3 --kernel-trace --stats of bench.py --steps 5 --warmup 2 --no-cpu-baseline [args]
#   pmc[:args]       the PMC passes (SQ instruction counters, FETCH_SIZE, WRITE_SIZE, the VALU instruction classes) at the bench's own batch size: the kernels
#                    the bench line runs (launch_autoc2 picks by the number of wavefronts)
#   ab:<rounds>[:args]  alternate flac_amd/lib (A) and build/alt_lib (B) engines, scripts/gpu_ab.sh
#   sh:<file>        run another script of scripts/
#   ubench           flac_amd/lib/ubench_cycles (scripts/ubench_cycles.hip, built here): shader cycles per wavefront-instruction per class -> ubench_cycles.json
set -u
TAG=${1:-visit}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# a box whose GPU does not answer a trivial kernel within a minute, or whose test step fails, gets no further steps: a visit once
# spent 25 GPU-minutes in the timeouts of counter passes behind a memory fault in the first process
timeout 90 python -c "import torch; print('gpu ok', float(torch.ones(4, device='cuda').sum()))" || { echo "GPU health check failed: visit abandoned"; exit 3; }
i=0
for step in "$@"; do
  i=$((i+1))
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}; arg=${arg//,/ }
  case $kind in
    env) if [ "${arg:0:1}" = "-" ]; then unset ${arg:1}; else export "$arg"; fi; echo "[$i] env $arg" ;;
    tests)
      if [ -n "$arg" ]; then FLACGPU_POISON=1 timeout 1500 python -m pytest tests -x -q -m gpu -k "$arg" > $OUT/pytest_$i.log 2>&1
      else FLACGPU_POISON=1 timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_$i.log 2>&1; fi
      rc=$?; echo "[$i] pytest rc=$rc"; tail -4 $OUT/pytest_$i.log
      if [ $rc -ne 0 ]; then echo "test step failed: visit abandoned"; exit 4; fi ;;
    bench)
      timeout 900 python bench.py $arg > $OUT/bench_$i.json 2> $OUT/bench_$i.err; echo "[$i] bench $arg rc=$?"; cat $OUT/bench_$i.json; tail -2 $OUT/bench_$i.err ;;
    prof)
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof$i -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline $arg > $OUT/prof_bench_$i.json 2> $OUT/prof_$i.err
      echo "[$i] prof rc=$?"
      DB=$(ls $OUT/prof$i/*.db 2>/dev/null | head -1)
      [ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats_$i.txt && cat $OUT/kernel_stats_$i.txt
      rm -rf $OUT/prof$i ;;
    pmc)
      j=0
      for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
                 "SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT" \
                 "FETCH_SIZE" "WRITE_SIZE" \
                 "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"; do
        j=$((j+1))
        timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$j -o p$j -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --no-clock $arg > $OUT/pmc$j.json 2> $OUT/pmc$j.err
        echo "[$i] pmc pass $j rc=$? : $SET"
        DB=$(ls $OUT/pmc$j/*.db 2>/dev/null | head -1)
        [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters_$i.txt
        rm -rf $OUT/pmc$j
      done
      cat $OUT/pmc_counters_$i.txt ;;
    abn) r=${arg%% *}; rest=""; [ "$r" != "$arg" ] && rest=${arg#* }; bash scripts/gpu_abn.sh $r $rest 2>&1 | tee $OUT/abn_$i.txt ;;
    ab) r=${arg%% *}; rest=""; [ "$r" != "$arg" ] && rest=${arg#* }; bash scripts/gpu_ab.sh $r $rest 2>&1 | tee $OUT/ab_$i.txt ;;
    sh) bash scripts/$arg 2>&1 | tee $OUT/sh_$i.txt ;;
    ubench) timeout 300 flac_amd/lib/ubench_cycles $OUT/ubench_cycles.json 2>&1 | tee $OUT/ubench_cycles.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
