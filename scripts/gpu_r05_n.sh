#!/bin/bash
mkdir -p gpurun_out/r05_n
export TMPDIR=/tmp
cat > /tmp/dbg3.py <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch, flac_amd, signals
from oracle import pyoracle as po
def run(ch, nfr, level, bps, fam="music", bs=1152, scale=None, **kw):
    pcm = signals.FAMILIES[fam](bs * nfr + 77, ch, bps)
    if scale == "full":
        rng = np.random.default_rng(3); pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=pcm.shape, dtype=np.int64).astype(np.int32)
    if scale == "square":
        pcm = np.where((np.arange(pcm.shape[0])[:, None] // 3) % 2 == 0, (1 << (bps - 1)) - 1, -(1 << (bps - 1))).astype(np.int32) * np.ones((1, ch), np.int32)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, level, **kw), device=0, max_batch_frames=nfr + 1)
    data, fb = eng.encode(pcm); k = eng.last_batch_kernels(); eng.close()
    o = po.oracle_encode(pcm, bps, 48000, level)
    print("ch", ch, "frames", nfr, "level", level, "bps", bps, fam, scale, "ok" if data == o["data"] else "DIFFERS", sorted(x for x in k if "prep" in x or "eval" in x), flush=True)
for bps in (24, 20, 17, 18):
    for level in (0, 1, 2):
        run(2, 40, level, bps)
        run(1, 40, level, bps)
run(2, 30, 0, 24, scale="full"); run(2, 30, 2, 24, scale="full"); run(2, 30, 2, 24, scale="square"); run(1, 30, 0, 24, scale="square")
run(2, 30, 2, 24, "wasted"); run(2, 30, 1, 24, "mixed"); run(6, 12, 0, 24); run(2, 30, 2, 24, "quiet"); run(2, 30, 0, 24, "sine"); run(2, 30, 2, 24, "constant"); run(2, 30, 2, 24, "silence")
PY
timeout 300 python /tmp/dbg3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_n/dbg.txt
timeout 600 python scripts/matrix_rate.py 8192 2>&1 | grep -v amdgpu.ids | grep "24-bit" | tee gpurun_out/r05_n/matrix_rate_24.txt
