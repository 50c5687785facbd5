#!/usr/bin/env python3
"""Side measurement: per-kernel times of one resident batch of independent channels (mono, stereo without mid/side, 5.1) at -8 with
other block sizes (prep4_kernel<., NW, CH>, round 6).  usage: chan_block_rate.py [M samples per batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

TOTAL = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) << 20
for ch, kw, name in ((1, {}, "mono"), (2, dict(mid_side=0), "stereo, no mid/side"), (6, {}, "5.1")):
    for N in (4096, 1152, 2304, 4608, 8192, 1024):
        nf = TOTAL // N
        base = signals.music(64 * N, ch, 16, seed=5)
        pcm = np.tile(base, ((nf + 63) // 64, 1))[: nf * N]
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, 16, 48000, 8, blocksize=N, streamable_subset=0, **kw), device=0, max_batch_frames=nf)
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
        k = sorted(x for x in eng.last_batch_kernels() if x.startswith("prep"))
        print("%-20s -b %-5d %6.3f ms per %d samples = %7.1f M samples/s  %s  %s" % (name, N, tot, nf * N, nf * N / tot / 1e3, {k_: round(v, 3) for k_, v in ms.items()}, " ".join(k)))
        eng.close()
        del d_pcm, d_out
