#!/usr/bin/env python3
"""Side measurement: per-kernel times of one resident batch of 16-bit stereo at -8 with other maximum LPC orders (-l 12 is the
preset's; the reference's own test matrix uses -l 16, FLAC__MAX_LPC_ORDER is 32) and other block sizes.  usage: order_rate.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, kw in (("-8 (-l 12)", {}), ("-8 -l 13", dict(max_lpc_order=13, streamable_subset=0)), ("-8 -l 16", dict(max_lpc_order=16, streamable_subset=0)), ("-8 -l 32", dict(max_lpc_order=32, streamable_subset=0)),
                 ("-8 -b 2304", dict(blocksize=2304)), ("-8 -b 1152", dict(blocksize=1152)), ("-8 -b 8192 --lax", dict(blocksize=8192, streamable_subset=0)),
                 ("-8 -b 4608", dict(blocksize=4608)), ("-8 -b 2048", dict(blocksize=2048)), ("-8 -b 1024", dict(blocksize=1024))):
    N = kw.get("blocksize", 4096)
    nf = NF * 4096 // N
    base = signals.music(64 * N, 2, 16, seed=5)
    pcm = np.tile(base, ((nf + 63) // 64, 1))[: nf * N]
    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8, **kw), device=0, max_batch_frames=nf)
    d_pcm = torch.from_numpy(pcm).cuda()
    cap = eng.max_output_bytes(nf)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_fb = torch.empty(nf, dtype=torch.int32, device="cuda")
    d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        eng.encode_device(d_pcm.data_ptr(), nf, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
    torch.cuda.synchronize()
    ms = eng.last_phase_ms()
    tot = sum(ms.values())
    k = sorted(x for x in eng.last_batch_kernels() if "<" not in x)
    print("%-18s %7.3f ms per %d samples = %8.1f M samples/s  %s  %s" % (name, tot, nf * N, nf * N / tot / 1e3, {k_: round(v, 3) for k_, v in ms.items()}, " ".join(k)))
    eng.close()
