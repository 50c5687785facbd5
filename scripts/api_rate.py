#!/usr/bin/env python3
"""End-to-end rate of the drop-in encoder API (libFLACgpu.so: FLAC__stream_encoder_* from host memory, write callback to
memory) next to the reference library driven the same way.  usage: api_rate.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import signals
import flac_api as fa

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = signals.music(512 * 4096, 2, 16, seed=3)
pcm = np.tile(base, ((NF + 511) // 512, 1))[: NF * 4096]
for which in ("gpu", "ref"):
    for md5 in (1, 0):
        if which == "ref" and NF > 1024:
            sub = pcm[: 1024 * 4096]
        else:
            sub = pcm
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            data, _ = fa.encode(which, sub, 16, 44100, 8, chunk=1 << 20, settings=(("set_do_md5", md5),))
            best = min(best, time.perf_counter() - t0)
        print("%s md5=%d: %8.1f M samples/s  (%d samples, %d bytes)" % (which, md5, sub.shape[0] / best / 1e6, sub.shape[0], len(data)), flush=True)
