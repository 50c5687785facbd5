#!/bin/bash
# after an evidence visit (scripts/gpu_visit.sh <tag> tests bench ubench prof pmc prof:... pmc:... for the five workloads in the order
# level8 level5 level0 white8 hires8): the summaries into profiles/, pmc_traffic.json regenerated.  usage: evidence_collect.sh <tag> <name>
set -e
TAG=$1; NAME=${2:-$1}
G=gpurun_out/$TAG
i=4
for w in level8 level5 level0 white8 hires8; do
  cp $G/kernel_stats_$i.txt profiles/${NAME}_kernel_stats_$w.txt
  cp $G/pmc_counters_$((i+1)).txt profiles/${NAME}_pmc_counters_$w.txt
  i=$((i+2))
done
cp $G/bench_2.json profiles/${NAME}_bench_before_counters.json
cp $G/ubench_cycles.json profiles/ubench_cycles.json
cp $G/ubench_cycles.txt profiles/${NAME}_ubench_cycles.txt
tail -3 $G/pytest_1.log > profiles/${NAME}_pytest_gpu.txt
python scripts/pmc_to_traffic.py $NAME level8=profiles/${NAME}_pmc_counters_level8.txt:262144:4096 level5=profiles/${NAME}_pmc_counters_level5.txt:65536:4096 \
  level0=profiles/${NAME}_pmc_counters_level0.txt:262144:1152 white8=profiles/${NAME}_pmc_counters_white8.txt:65536:4096 hires8=profiles/${NAME}_pmc_counters_hires8.txt:65536:4096
