#!/usr/bin/env python3
"""PCIe-inclusive encode rate: host buffers through flacgpu_encode_batch (int32 block) and flacgpu_encode_batch_raw
(16-bit little-endian file bytes, staged on the device), pinned staging memory.  Never bench.py's `value`."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_amd  # noqa: E402
import signals  # noqa: E402
from rawfmt import to_raw  # noqa: E402

NF, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 4096
eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8), device=0, max_batch_frames=NF)
lib = eng.lib
lib.flacgpu_alloc_pinned.restype = C.c_void_p
lib.flacgpu_alloc_pinned.argtypes = [C.c_size_t]
base = signals.music(512 * N, 2, 16, seed=3)
pcm = np.tile(base, ((NF + 511) // 512, 1))[: NF * N]
raw = to_raw(pcm, 16)


def pinned(arr):
    p = lib.flacgpu_alloc_pinned(arr.nbytes)
    buf = (C.c_uint8 * arr.nbytes).from_address(p)
    np.frombuffer(buf, dtype=np.uint8)[:] = arr.view(np.uint8).reshape(-1)
    return p


cap = eng.max_output_bytes(NF)
out_p = lib.flacgpu_alloc_pinned(cap)
fb = np.empty(NF, dtype=np.uint32)
p32, p16 = pinned(pcm), pinned(raw)
fmt = flac_amd.raw_format(16)
for name, call in (("int32 host block (flacgpu_encode_batch)", lambda: lib.flacgpu_encode_batch(eng.ctx, p32, NF, 0, 0, None, out_p, cap, fb.ctypes.data)),
                   ("16-bit file bytes (flacgpu_encode_batch_raw)", lambda: lib.flacgpu_encode_batch_raw(eng.ctx, p16, C.byref(fmt), NF, 0, 0, None, out_p, cap, fb.ctypes.data))):
    call()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        r = call()
        assert r > 0
    dt = (time.perf_counter() - t0) / reps
    print("%-48s %7.2f ms per %d frames  %8.1f M samples/s (PCIe inclusive, D2H of the frames included)" % (name, dt * 1e3, NF, NF * N / dt / 1e6))

# The asynchronous entry: ONE engine, ONE host thread, up to FLACGPU_ASYNC_SLOTS batches in flight -- the input copy of batch k+1
# and the read-back of batch k-1 run beside the kernels of batch k (flacgpu_submit_batch_raw / flacgpu_collect)
DEPTH = 4
srcs = [p16] + [pinned(raw) for _ in range(DEPTH - 1)]
outs = [out_p] + [lib.flacgpu_alloc_pinned(cap) for _ in range(DEPTH - 1)]
fbs = [np.empty(NF, dtype=np.uint32) for _ in range(DEPTH)]
for depth in (1, 2, 3, 4):
    def run(nb):
        sub = col = 0
        tot = 0
        while col < nb:
            while sub < nb and sub - col < depth:
                k = sub % DEPTH
                r = lib.flacgpu_submit_batch_raw(eng.ctx, srcs[k], C.byref(fmt), NF, sub * NF, 0, None, outs[k], cap, fbs[k].ctypes.data)
                assert r == 0, r
                sub += 1
            r = lib.flacgpu_collect(eng.ctx)
            assert r > 0, r
            tot += r
            col += 1
        return tot
    run(2)
    reps = 12
    t0 = time.perf_counter()
    run(reps)
    dt = (time.perf_counter() - t0) / reps
    print("%-48s %7.2f ms per %d frames  %8.1f M samples/s (one engine, one host thread)" % ("submit/collect, %d batch(es) in flight" % depth, dt * 1e3, NF, NF * N / dt / 1e6))

# Two engines, two host threads: each call is synchronous on its own HIP stream (H2D copy -> kernels -> D2H copy), so two
# callers overlap one another's copies and kernels -- the pipelined use of the ABI (a corpus encoder keeps two batches in flight).
import threading  # noqa: E402
eng2 = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8), device=0, max_batch_frames=NF)
out2_p = lib.flacgpu_alloc_pinned(cap)
fb2 = np.empty(NF, dtype=np.uint32)
p16b = pinned(raw)
reps = 6


def worker(e, src, outp, fbb):
    for _ in range(reps):
        r = lib.flacgpu_encode_batch_raw(e.ctx, src, C.byref(fmt), NF, 0, 0, None, outp, cap, fbb.ctypes.data)
        assert r > 0


worker(eng2, p16b, out2_p, fb2)
t0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=a) for a in ((eng, p16, out_p, fb), (eng2, p16b, out2_p, fb2))]
for t in ths:
    t.start()
for t in ths:
    t.join()
dt = (time.perf_counter() - t0) / (2 * reps)
print("%-48s %7.2f ms per %d frames  %8.1f M samples/s (two engines, two host threads)" % ("16-bit file bytes, 2 batches in flight", dt * 1e3, NF, NF * N / dt / 1e6))
eng2.close()
eng.close()
