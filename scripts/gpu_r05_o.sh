#!/bin/bash
# independent channels, a wavefront per window-job set: tests + A/B of scripts/chan_rate.py (mono at 16384 frames is the size it is for)
mkdir -p gpurun_out/r05_o
export TMPDIR=/tmp
O=gpurun_out/r05_o
timeout 900 python -m pytest tests/test_headline_selection_gpu.py -x -q -m gpu -k "window_job_set or independent_channels or off_the_headline or edge_signals" 2>&1 | tail -5 | tee $O/pytest.log
for nf in 16384 12288 20480; do
  for v in 0 2; do
    echo "== mono, $nf frames, FLACGPU_AUTOC3_IND_SETS=$v" | tee -a $O/chan_rate_ab.txt
    CHAN_ONLY=mono FLACGPU_AUTOC3_IND_SETS=$v timeout 300 python scripts/chan_rate.py $nf 2>&1 | grep -v amdgpu.ids | tee -a $O/chan_rate_ab.txt
  done
done
echo "== all layouts, 16384 frames, default" | tee -a $O/chan_rate_ab.txt
timeout 600 python scripts/chan_rate.py 16384 2>&1 | grep -v amdgpu.ids | tee -a $O/chan_rate_ab.txt
echo "== stereo without mid/side + 5.1 at 4096 frames (stereo: 128 groups; 5.1: 384 groups -> not by sets), default vs 0" | tee -a $O/chan_rate_ab.txt
for v in 0 2; do FLACGPU_AUTOC3_IND_SETS=$v timeout 300 python scripts/chan_rate.py 5120 2>&1 | grep -v amdgpu.ids | tee -a $O/chan_rate_ab.txt; done
