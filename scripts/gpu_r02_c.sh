#!/bin/bash
# round 2, visit C: everything after the fused compaction / grouped autocorrelation / prep3 sector stores / log table in LDS /
# verify with lane-interleaved decode + compare kernel: full GPU suite, bench, A/B of the two main changes, PMC traffic.
set -u
TAG=${1:-r02_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],d['kernel_ms'])
print('verify',d.get('device_verify'),'ok',d['verified']['ok'])
for k in ('white_noise','level5'): print(k,d[k]['value'],d[k]['kernel_ms'],d[k]['verified_ok'])
PY
tail -3 $OUT/bench.err
for v in "FLACGPU_NO_FUSED_COMPACT=1" "FLACGPU_AUTOC2_UNGROUPED=1" "X=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --no-verify --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])"
done
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc pass $i rc=$? : $SET"
  DB=$(ls $OUT/pmc$i/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
  rm -rf $OUT/pmc$i
done
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc_counters.txt | grep -v "copyBuffer\|elementwise"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify > $OUT/prof_bench.json 2> $OUT/prof.err
DB=$(ls $OUT/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt && head -14 $OUT/kernel_stats.txt
rm -rf $OUT/prof
