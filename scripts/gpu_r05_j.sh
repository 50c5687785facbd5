#!/bin/bash
mkdir -p gpurun_out/r05_j
export TMPDIR=/tmp
cat > /tmp/dbg2.py <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch, flac_amd, signals
from oracle import pyoracle as po
def run(ch, nfr, order, bps=16, bs=4096, fam="music", **kw):
    pcm = signals.FAMILIES[fam](bs * nfr + 55, ch, bps)
    ekw = dict(max_lpc_order=order, streamable_subset=0, blocksize=bs, **kw)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, 8, **ekw), device=0, max_batch_frames=nfr + 1)
    data, fb = eng.encode(pcm); k = eng.last_batch_kernels(); eng.close()
    okw = dict(max_lpc_order=order, blocksize=bs)
    if "mid_side" in kw: okw.update(mid_side=kw["mid_side"], loose=kw.get("loose_mid_side", 0))
    o = po.oracle_encode(pcm, bps, 48000, 8, **okw)
    print("ch", ch, "frames", nfr, "order", order, "bps", bps, "bs", bs, fam, "ok" if data == o["data"] else "DIFFERS", sorted(x for x in k if "autoc" in x or "eval" in x), flush=True)
for order in (16, 17, 20, 24, 25, 31, 32):
    run(2, 40, order)
run(1, 70, 16); run(6, 12, 32); run(2, 30, 16, 24); run(2, 20, 32, 16, 4608); run(2, 25, 16, 16, 1152); run(2, 20, 16, 16, 4096, "sine"); run(2, 20, 32, 16, 4096, "wasted")
run(2, 40, 16, mid_side=0); run(2, 40, 20, mid_side=1, loose_mid_side=1); run(2, 33, 16, 16, 576)
PY
FLACGPU_AUTOC2=1 FLACGPU_SYNC_DEBUG=1 timeout 300 python /tmp/dbg2.py 2>&1 | grep -v "\.\.\. ok" | grep -v amdgpu.ids | tee gpurun_out/r05_j/dbg.txt
timeout 300 python scripts/order_rate.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_j/order_rate.txt
