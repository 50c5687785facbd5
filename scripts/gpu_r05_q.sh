#!/bin/bash
# timing events without the system fence, results written in place: A/B of bench lines (-0, -5, -8) + timelines + tests
mkdir -p gpurun_out/r05_q
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_q
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "verify or golden or async or submit or device or tail or short" 2>&1 | tail -4 | tee $O/pytest.log
for lv in 0 5 8; do
  for rep in 1 2; do
    for mode in old new; do
      if [ $mode = old ]; then export FLACGPU_EVENT_FENCE=1 FLACGPU_COPY_RESULTS=1; else unset FLACGPU_EVENT_FENCE FLACGPU_COPY_RESULTS; fi
      timeout 300 python bench.py --level $lv --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify --no-clock 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level $lv $mode', d['value'], d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    done
  done
done
unset FLACGPU_EVENT_FENCE FLACGPU_COPY_RESULTS
for lv in 0 5; do
  timeout 300 rocprofv3 --kernel-trace -d $O/kt$lv -o kt -- python bench.py --level $lv --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-verify --no-clock > $O/bench_$lv.json 2> $O/bench_$lv.err
  DB=$(ls $O/kt$lv/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_timeline.py $DB 30 > $O/timeline_$lv.txt
  rm -rf $O/kt$lv
  head -24 $O/timeline_$lv.txt
done
