#!/bin/bash
# is it the batch size or the duration?  -8 at 16384 frames with 20 / 200 / 1000 timed steps against 262144 frames with 3 / 10
TAG=$1; R=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; steps=$1; shift; warm=$1; shift; bargs=$1; shift
  env "$@" timeout 400 python bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-extras --no-api --no-decode --no-verify $bargs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('clock',{}).get('mhz_mean'), d.get('clock',{}).get('mhz_p10'))"; }
for i in $(seq $R); do
  run l8_16k_s20 20 2 "--level 8" X=1
  run l8_16k_s200 200 50 "--level 8" X=1
  run l8_16k_s1000 1000 200 "--level 8" X=1
  run l8_64k_s50 50 10 "--level 8 --frames 65536" X=1
  run l8_256k_s3 3 2 "--level 8 --frames 262144" X=1
  run l8_256k_s10 10 2 "--level 8 --frames 262144" X=1
  run l5_16k_s20 20 2 "--level 5" X=1
  run l5_16k_s1000 1000 200 "--level 5" X=1
  run l5_256k_s10 10 2 "--level 5 --frames 262144" X=1
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
