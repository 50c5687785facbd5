#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_mono}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for who in mono "stereo -8"; do
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  CHAN_ONLY="$who" timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc -o p -- python scripts/chan_rate.py 4096 > /dev/null 2> $OUT/pmc.err
  DB=$(ls $OUT/pmc/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB | grep autoc | sed "s/^/[$who] /" >> $OUT/mono_pmc.txt
  rm -rf $OUT/pmc
done; done
cat $OUT/mono_pmc.txt | cut -c1-140
