#!/usr/bin/env python3
"""Side measurement: does a frame count that is no multiple of the kernels' group sizes (16 frames, 64 subframes) cost more per frame?
us per frame of one resident batch at -8 / -5, stereo and mono, round and odd counts.  usage: odd_count_rate.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

N = 4096
for ch, level, counts in ((2, 8, (32768, 32768 + 7, 32768 + 33, 32768 - 1)), (2, 5, (32768, 32768 + 7, 32768 - 1)), (1, 8, (16384, 16384 + 37, 16384 - 1, 10000, 10001)),
                          (6, 8, (4096, 4096 + 5, 4096 - 1)), (2, 0, (65536, 65536 + 7))):
    for nf in counts:
        bs = 1152 if level < 3 else N
        base = signals.music(64 * bs, ch, 16, seed=5)
        pcm = np.tile(base, ((nf + 63) // 64, 1))[: nf * bs]
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, 16, 48000, level), device=0, max_batch_frames=nf)
        d_pcm = torch.from_numpy(pcm).cuda()
        cap = eng.max_output_bytes(nf)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_fb = torch.empty(nf, dtype=torch.int32, device="cuda")
        d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
        for _ in range(4):
            eng.encode_device(d_pcm.data_ptr(), nf, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
        torch.cuda.synchronize()
        ms = eng.last_phase_ms()
        tot = sum(ms.values())
        print("%d ch -%d %7d frames  %8.3f ms = %7.4f us per frame  %s" % (ch, level, nf, tot, tot * 1e3 / nf, {k: round(v * 1e3 / nf, 4) for k, v in ms.items() if v}))
        eng.close()
        del d_pcm, d_out
