#!/bin/bash
# batch-size survey, second part: 64 Ki .. 256 Ki frames per step, phase timing never.  usage: scripts/gpu_r06_v.sh <tag> [rounds]
TAG=$1; R=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
free -g | head -2
run() { name=$1; shift; steps=$1; shift; bargs=$1; shift
  SECONDS=0; env "$@" timeout 400 python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-extras --no-clock --no-api --no-decode $bargs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('verified',{}).get('ok'), d.get('device_verify',{}).get('status'))"; echo "$name wall $SECONDS s"; }
for i in $(seq $R); do
  run l8_16k 20 "--level 8" X=1
  run l8_16k_t0 20 "--level 8 --timing-every 0" X=1
  run l8_64k 6 "--level 8 --frames 65536" X=1
  run l8_64k_t0 6 "--level 8 --frames 65536 --timing-every 0" X=1
  run l8_128k 4 "--level 8 --frames 131072" X=1
  run l8_256k 3 "--level 8 --frames 262144" X=1
  run l5_128k 4 "--level 5 --frames 131072" X=1
  run l5_16k_t0 20 "--level 5 --timing-every 0" X=1
  run hires_64k 4 "--hires --frames 65536" X=1
  run hires_128k 3 "--hires --frames 131072" X=1
  run l0_16k_t0 20 "--level 0 --timing-every 0" X=1
  FLACGPU_DEBUG_TIMING=1 python bench.py --level 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-clock --no-api --no-decode --no-verify 2>&1 | grep flacgpu | tail -8
  FLACGPU_DEBUG_TIMING=1 python bench.py --level 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-clock --no-api --no-decode --no-verify 2>&1 | grep flacgpu | tail -8
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
