#!/usr/bin/env python3
"""Side measurement: per-kernel times of one resident batch for other channel layouts (mono, stereo without mid/side,
5.1) at -8.  usage: chan_rate.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402

NF, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 4096
ONLY = os.environ.get("CHAN_ONLY")             # e.g. CHAN_ONLY=mono under a profiler
for ch, kw, name in ((1, {}, "mono"), (2, dict(mid_side=0), "stereo, no mid/side"), (2, {}, "stereo -8"), (6, {}, "5.1")):
    if ONLY and not name.startswith(ONLY):
        continue
    base = signals.music(64 * N, ch, 16, seed=5)
    pcm = np.tile(base, ((NF + 63) // 64, 1))[: NF * N]
    eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, 16, 48000, 8, **kw), device=0, max_batch_frames=NF)
    d_pcm = torch.from_numpy(pcm).cuda()
    cap = eng.max_output_bytes(NF)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_fb = torch.empty(NF, dtype=torch.int32, device="cuda")
    d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        eng.encode_device(d_pcm.data_ptr(), NF, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
    torch.cuda.synchronize()
    ms = eng.last_phase_ms()
    tot = sum(ms.values())
    print("%-20s %6.3f ms per %d frames = %7.1f M samples/s (%6.1f M channel-samples/s)  %s" % (name, tot, NF, NF * N / tot / 1e3, NF * N * ch / tot / 1e3, {k: round(v, 3) for k, v in ms.items()}))
    eng.close()
