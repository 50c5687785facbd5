#!/bin/bash
# round 2, visit A: GPU tests (incl. the log pin and the corpus job), the new bench line, the multi-rank pipeline on one
# rank, BASELINE config 5 for real (10 h), and a kernel-trace profile of the starting point.
set -u
TAG=${1:-r02_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lscpu | head -25 > $OUT/lscpu.txt 2>&1
free -g >> $OUT/lscpu.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $OUT/bench_plain.json 2> $OUT/bench_plain.err; echo "plain rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_plain.json'));print(d['value'],d['ms_per_step'],d['kernel_ms'])"
timeout 300 python bench.py --force-dist --no-cpu-baseline --steps 20 > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "forcedist rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_forcedist.json'));print(d['value'],d['ms_per_step'],d['gather'],d.get('verified',{}).get('ok'))"; tail -3 $OUT/bench_forcedist.err
timeout 1200 python scripts/run_corpus.py --tag $TAG --hours 10 > $OUT/corpus_stdout.json 2> $OUT/corpus.err; echo "corpus rc=$?"; cat $OUT/corpus_stdout.json; tail -5 $OUT/corpus.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify > $OUT/prof_bench.json 2> $OUT/prof.err
echo "prof rc=$?"
DB=$(ls $OUT/prof/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
rm -rf $OUT/prof
