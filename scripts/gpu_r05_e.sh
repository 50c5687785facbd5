#!/bin/bash
mkdir -p gpurun_out/r05_e
export TMPDIR=/tmp
cat > /tmp/dbg.py <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch, flac_amd, signals
from oracle import pyoracle as po
def run(ch, nfr, level=8, bps=16):
    pcm = signals.music(4096 * nfr, ch, bps, seed=3)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, level, mid_side=0), device=0, max_batch_frames=nfr)
    data, fb = eng.encode(pcm); k = eng.last_batch_kernels(); eng.close()
    o = po.oracle_encode(pcm, bps, 48000, level, mid_side=0)
    print("ch", ch, "frames", nfr, "level", level, "bps", bps, "ok" if data == o["data"] else "DIFFERS", sorted(x for x in k if "prep" in x or "autoc" in x), flush=True)
for ch, nfr in ((1, 20), (1, 130), (2, 70), (6, 30), (1, 700), (8, 9), (5, 14), (3, 25), (7, 11), (4, 17)):
    run(ch, nfr)
run(1, 130, 5); run(2, 40, 8, 24)
PY
for env in "FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1"; do
  echo "=== $env" >> gpurun_out/r05_e/dbg.txt
  env $env FLACGPU_SYNC_DEBUG=1 timeout 200 python /tmp/dbg.py >> gpurun_out/r05_e/dbg.txt 2>&1
done
grep -v "\.\.\. ok" gpurun_out/r05_e/dbg.txt | tail -30
timeout 300 python scripts/chan_rate.py 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_e/chan_rate.txt
