#!/bin/bash
# the bench line with the round-6 defaults (262144 frames per step), the large-batch test, the multi-rank path at world size 1
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SECONDS=0
timeout 900 python -m pytest tests/test_large_batch_gpu.py -x -q -m gpu > $OUT/pytest_large.log 2>&1; echo "pytest large rc=$? ($SECONDS s)"; tail -5 $OUT/pytest_large.log
SECONDS=0
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? ($SECONDS s)"; cat $OUT/bench.json; tail -3 $OUT/bench.err
SECONDS=0
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 timeout 900 python bench.py --force-dist --no-cpu-baseline --no-extras > $OUT/bench_dist.json 2> $OUT/bench_dist.err; echo "bench dist rc=$? ($SECONDS s)"; cat $OUT/bench_dist.json; tail -3 $OUT/bench_dist.err
SECONDS=0
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $OUT/pytest_bench.log 2>&1; echo "pytest bench rc=$? ($SECONDS s)"; tail -5 $OUT/pytest_bench.log
