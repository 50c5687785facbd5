#!/bin/bash
# round 2, visit E: the drop-in test on the GPU, single-stream API and CLI rates, full suite, the bench line of the committed
# state with its kernel-trace and PMC passes.
set -u
TAG=${1:-r02_e}
OUT=gpurun_out/$TAG
mkdir -p $OUT build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
gcc -O2 -Iinclude scripts/api_rate.c -o build/api_rate -Lflac_amd/lib -lFLACgpu -lm -Wl,-rpath,$PWD/flac_amd/lib && {
  for md5 in 0 1; do ./build/api_rate 32768 8 $md5 2>&1 | tail -2; done
  FLACGPU_BATCH_FRAMES=16384 ./build/api_rate 32768 8 0 2>&1 | tail -1
  ./build/api_rate 32768 5 0 2>&1 | tail -1
} | tee $OUT/api_rate.txt
timeout 600 python scripts/cli_rate.py 30 2>&1 | tee $OUT/cli_rate.txt
timeout 900 python bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -2 $OUT/bench.err
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
