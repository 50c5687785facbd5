#!/bin/bash
# batch-size and stream-overlap survey: -8 / -5 / -0 / 96 kHz 24-bit at 16384, 32768, 65536 frames per step; -5 with two sub-batches;
# phase timing never.  usage: scripts/gpu_r06_u.sh <tag> [rounds]
TAG=$1; R=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; steps=$1; shift; bargs=$1; shift
  env "$@" timeout 300 python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-extras --no-clock --no-api --no-decode $bargs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('verified',{}).get('ok'), d.get('device_verify',{}).get('status'))"; }
for i in $(seq $R); do
  run l8_16k 20 "--level 8" X=1
  run l8_16k_t0 20 "--level 8 --timing-every 0" X=1
  run l8_32k 10 "--level 8 --frames 32768" X=1
  run l8_64k 6 "--level 8 --frames 65536" X=1
  run l5_16k 20 "--level 5" X=1
  run l5_16k_sub2 20 "--level 5" FLACGPU_SUBBATCHES=2
  run l5_16k_t0 20 "--level 5 --timing-every 0" X=1
  run l5_32k 10 "--level 5 --frames 32768" X=1
  run l5_64k 6 "--level 5 --frames 65536" X=1
  run l0_16k 20 "--level 0" X=1
  run l0_64k 10 "--level 0 --frames 65536" X=1
  run l0_256k 6 "--level 0 --frames 262144" X=1
  run hires_16k 10 "--hires" X=1
  run hires_32k 6 "--hires --frames 32768" X=1
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
