#!/bin/bash
# round 2, visit B: GPU verify tests, bench with the device self check, PMC baseline of the round's starting kernels,
# phase stamps of pack2 / eval, cgroup CPU quota of the box.
set -u
TAG=${1:-r02_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( cat /sys/fs/cgroup/cpu.max; nproc; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us ) > $OUT/cpus.txt 2>&1
cat $OUT/cpus.txt
timeout 1200 python -m pytest tests/test_verify_gpu.py tests/test_stream_encoder_api.py tests/test_flac_cli.py -x -q -m gpu > $OUT/pytest_verify.log 2>&1; echo "pytest verify rc=$?"; tail -15 $OUT/pytest_verify.log
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > $OUT/bench_verify.json 2> $OUT/bench_verify.err; echo "bench rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_verify.json'));print(d['value'],d['kernel_ms'],d.get('device_verify'),d['verified']['ok'])"; tail -3 $OUT/bench_verify.err
FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-verify --steps 2 --warmup 1 > $OUT/bench_dbg.json 2> $OUT/bench_dbg.err; grep flacgpu $OUT/bench_dbg.err | tail -12
FLACGPU_DEBUG_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-verify --steps 2 --warmup 1 --level 5 > $OUT/bench_dbg5.json 2> $OUT/bench_dbg5.err; grep flacgpu $OUT/bench_dbg5.err | tail -12
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/pmc$i -o p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --frames 4096 > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc pass $i rc=$? : $SET"
  DB=$(ls $OUT/pmc$i/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_pmc.py $DB >> $OUT/pmc_counters.txt
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_counters.txt | head -120
