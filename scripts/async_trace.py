#!/usr/bin/env python3
"""where the host's time goes in flacgpu_submit_batch_raw / flacgpu_collect (development aid)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import flac_amd, signals
from rawfmt import to_raw
NF, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 4096
eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8), device=0, max_batch_frames=NF)
lib = eng.lib
lib.flacgpu_alloc_pinned.restype = C.c_void_p; lib.flacgpu_alloc_pinned.argtypes = [C.c_size_t]
base = signals.music(512 * N, 2, 16, seed=3)
raw = to_raw(np.tile(base, ((NF + 511) // 512, 1))[: NF * N], 16)
def pinned(arr):
    p = lib.flacgpu_alloc_pinned(arr.nbytes)
    np.frombuffer((C.c_uint8 * arr.nbytes).from_address(p), dtype=np.uint8)[:] = arr.view(np.uint8).reshape(-1)
    return p
cap = eng.max_output_bytes(NF)
D = 4
srcs = [pinned(raw) for _ in range(D)]; outs = [lib.flacgpu_alloc_pinned(cap) for _ in range(D)]; fbs = [np.empty(NF, dtype=np.uint32) for _ in range(D)]
fmt = flac_amd.raw_format(16)
for depth in (1, 4):
    for rep in range(2):
        sub = col = 0; nb = 10; ev = []
        t00 = time.perf_counter()
        while col < nb:
            while sub < nb and sub - col < depth:
                k = sub % D; t0 = time.perf_counter()
                r = lib.flacgpu_submit_batch_raw(eng.ctx, srcs[k], C.byref(fmt), NF, sub * NF, 0, None, outs[k], cap, fbs[k].ctypes.data); assert r == 0
                ev.append(("S%d" % sub, t0 - t00, time.perf_counter() - t0)); sub += 1
            t0 = time.perf_counter(); r = lib.flacgpu_collect(eng.ctx); assert r > 0
            ev.append(("C%d" % col, t0 - t00, time.perf_counter() - t0)); col += 1
        if rep: print("depth", depth, " ".join("%s@%.2f+%.2f" % (n, a * 1e3, d * 1e3) for n, a, d in ev), "total %.2f ms" % ((time.perf_counter() - t00) * 1e3))
