#!/usr/bin/env python3
"""Side measurement: 16-bit audio in a 24-bit container (8 wasted bits in every subframe: what a studio export of CD material looks like)
against plain 16-bit and plain 24-bit stereo at -8 / -5: per-kernel ms of one resident batch.  usage: wasted_rate.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

NF, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 4096
base16 = signals.music(64 * N, 2, 16, seed=5)
base24 = signals.music(64 * N, 2, 24, seed=5)
for level in (8, 5):
    for name, bps, base in (("16-bit", 16, base16), ("24-bit", 24, base24), ("16-bit audio in 24 bits (8 wasted)", 24, base16 << 8), ("20-bit audio in 24 bits (4 wasted)", 24, (base24 >> 4) << 4),
                            ("12-bit audio in 16 bits (4 wasted)", 16, (base16 >> 4) << 4)):
        pcm = np.tile(base, ((NF + 63) // 64, 1))[: NF * N].astype(np.int32)
        eng = flac_amd.FrameEngine(flac_amd.make_settings(2, bps, 48000, level), device=0, max_batch_frames=NF)
        d_pcm = torch.from_numpy(pcm).cuda()
        cap = eng.max_output_bytes(NF)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_fb = torch.empty(NF, dtype=torch.int32, device="cuda")
        d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
        for _ in range(6):
            eng.encode_device(d_pcm.data_ptr(), NF, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
        torch.cuda.synchronize()
        ms = eng.last_phase_ms()
        tot = sum(ms.values())
        k = sorted(x for x in eng.last_batch_kernels() if "<" not in x and x not in ("fused_output", "fo_place_kernel", "pack_plan_kernel", "model_kernel"))
        print("-%d %-36s %6.3f ms = %7.1f M samples/s  %s  %s" % (level, name, tot, NF * N / tot / 1e3, {k_: round(v, 3) for k_, v in ms.items() if v}, " ".join(k)))
        eng.close()
        del d_pcm, d_out
