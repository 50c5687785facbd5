#!/bin/bash
# round 5, visit a: the whole GPU suite on the new tree, the bench line, the multi-rank line at world 1 with its side figures, the corpus job as 120 tracks
mkdir -p gpurun_out/r05_a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r05_a/pytest.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a/bench.json 2> gpurun_out/r05_a/bench.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 timeout 300 python bench.py --gpus 1 --force-dist --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r05_a/bench_dist1.json 2> gpurun_out/r05_a/bench_dist1.err
timeout 300 python -m flac_amd.corpus --tracks 120 --hours 10 > gpurun_out/r05_a/corpus_tracks120.json 2> gpurun_out/r05_a/corpus_tracks120.err
timeout 300 python -m flac_amd.corpus --tracks 120 --hours 10 --input device --md5 host --md5-threads 4 > gpurun_out/r05_a/corpus_tracks120_r04way.json 2> gpurun_out/r05_a/corpus_tracks120_r04way.err
timeout 200 python -m flac_amd.corpus --tracks 1000 --hours 10 > gpurun_out/r05_a/corpus_tracks1000.json 2> gpurun_out/r05_a/corpus_tracks1000.err
df -h /dev/shm > gpurun_out/r05_a/shm.txt; nproc >> gpurun_out/r05_a/shm.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r05_a/shm.txt 2>&1; free -g >> gpurun_out/r05_a/shm.txt; lscpu | head -20 >> gpurun_out/r05_a/shm.txt
tail -3 gpurun_out/r05_a/pytest.log
