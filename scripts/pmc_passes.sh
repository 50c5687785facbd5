#!/bin/bash
# PMC counter passes for bench.py (each pass its own run; only --kernel-trace next to --pmc).
# usage: scripts/pmc_passes.sh <outdir-under-gpurun_out>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pass$i -o p$i -- $CMD > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "pass $i rc=$? : $SET"
done
ls -R $OUT | head -40
