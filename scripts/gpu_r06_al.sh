#!/bin/bash
# the final tree of round 6: both reference shell suites to their end on the drop-in, the soaks (sweep, adversarial, damaged streams,
# the one-kernel frame on adversarial signals), and how long the default bench.py takes on this box
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default bench.py: rc=$? in $SECONDS s" | tee $OUT/bench_seconds.txt
(time FLACGPU_SHELL_SUITE=full timeout 2400 python -m pytest tests/test_shell_suites_gpu.py -m gpu -q -x -s 2>&1 | tail -8) > $OUT/shell_suites_full.log 2>&1; cat $OUT/shell_suites_full.log
(time FLACGPU_TEST_SEEDS=800 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" 2>&1 | tail -4) > $OUT/sweep.log 2>&1; cat $OUT/sweep.log
(time FLACGPU_ADV_SEEDS=600 timeout 1500 python -m pytest tests/test_adversarial_gpu.py -m gpu -q 2>&1 | tail -4) > $OUT/adversarial.log 2>&1; cat $OUT/adversarial.log
(time FLACGPU_FF_SEEDS=1500 timeout 1500 python -m pytest tests/test_wide_ff_gpu.py -m gpu -q -k adversarial 2>&1 | tail -4) > $OUT/ff_adversarial.log 2>&1; cat $OUT/ff_adversarial.log
(time FLACGPU_SD_SEEDS=1500 timeout 1500 python -m pytest tests/test_stream_decode_gpu.py -m gpu -q -k "damaged" 2>&1 | tail -4) > $OUT/stream_decode.log 2>&1; cat $OUT/stream_decode.log
