#!/bin/bash
# One counter pass (development aid): the SQ instruction counters of bench.py's kernels at 4096 frames.  usage: scripts/gpu_pmc1.sh <tag> [kernel substring] [bench args]
TAG=$1; SUB=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 "$@" > $OUT/pmc.json 2> $OUT/pmc.err
DB=$(ls $OUT/pmc/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_pmc.py $DB "$SUB" | tee $OUT/pmc_counters.txt
rm -rf $OUT/pmc
