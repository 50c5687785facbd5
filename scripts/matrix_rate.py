#!/usr/bin/env python3
"""Side measurement: presets x input formats, per-kernel times of one resident batch (which kernels a shape gets, and where a
shape still runs on the general ones).  usage: matrix_rate.py [frames at 4096 samples]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
print("%-26s %-4s %9s %11s  %s" % ("input", "", "ms", "M samples/s", "kernel ms | kernels"))
for fmt, ch, bps, rate in (("16-bit stereo 44.1k", 2, 16, 44100), ("24-bit stereo 96k", 2, 24, 96000), ("16-bit mono 44.1k", 1, 16, 44100), ("16-bit 5.1 48k", 6, 16, 48000), ("24-bit mono 48k", 1, 24, 48000)):
    for level in (0, 2, 5, 8):
        N = 1152 if level < 3 else 4096
        nf = NF * 4096 // N
        base = signals.music(64 * N, ch, bps, seed=5)
        pcm = np.tile(base, ((nf + 63) // 64, 1))[: nf * N]
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, rate, level), device=0, max_batch_frames=nf)
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
        k = sorted(x for x in eng.last_batch_kernels() if "<" not in x and x not in ("fused_output", "fo_place_kernel", "pack_plan_kernel", "model_kernel"))
        print("%-26s -%d   %9.3f %11.1f  %s | %s" % (fmt, level, tot, nf * N / tot / 1e3, " ".join("%s %.3f" % (a, b) for a, b in ms.items() if b), " ".join(k)))
        eng.close()
        del d_pcm, d_out
