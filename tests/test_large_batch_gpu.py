"""-m gpu: batches far beyond the 16384 frames the suite and round 1-5's bench used (round 6: the bench's step is 262144 frames).

A batch is frames that do not know of each other (stream_encoder.c:3627-3744: one frame per thread-pool task), so ONE launch of F
frames must give, byte for byte, what the same frames give in batches of 16384 with their frame numbers carried along -- whatever F
is.  F is chosen so that the element index of the planar channels passes 2^31 and their byte offset 2^33 (140000 x 4 x 4096
samples), the worst-case slots pass 2^31 bytes, and the frame number reaches its 3-byte UTF-8 class: what a 32-bit index somewhere
in a kernel would break.  The small batches are the ones the oracle-pinned tests of this suite cover."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pcm(nframes, block, bps, seed):
    """a 512-frame clip of tests/signals.py's music repeated with gains and offsets, and every frame marked with its own number"""
    import signals
    base_frames = 512
    base = signals.music(base_frames * block, 2, bps, seed=seed, rate=96000 if bps > 16 else 44100).astype(np.float64)
    reps = (nframes + base_frames - 1) // base_frames
    lim = 1 << (bps - 1)
    out = np.empty((reps * base_frames * block, 2), dtype=np.int32)
    for r in range(reps):
        g = 1.0 - 0.05 * (r % 11)
        out[r * base_frames * block:(r + 1) * base_frames * block] = np.clip(np.rint(base * g) + (r % 7) - 3, -lim, lim - 1)
    out = out[: nframes * block]
    fr = out.reshape(nframes, block, 2)
    mark = (np.arange(nframes, dtype=np.int64) * 2654435761 % 61).astype(np.int32) - 30
    fr[:, 5, 0] = np.clip(fr[:, 5, 0] + mark, -lim, lim - 1)
    fr[:, block // 2, 1] = np.clip(fr[:, block // 2, 1] - mark, -lim, lim - 1)
    return out


@pytest.mark.parametrize("level,bps,nframes", [(8, 16, 140000), (5, 16, 140000), (0, 16, 300000), (8, 24, 70000)],
                         ids=["-8 16-bit 140000 frames", "-5 16-bit 140000 frames", "-0 16-bit 300000 frames", "-8 24-bit 70000 frames"])
def test_one_launch_equals_batches_of_16384(level, bps, nframes):
    import torch
    import flac_amd
    block = 1152 if level < 3 else 4096
    rate = 96000 if bps > 16 else 44100
    pcm_h = _pcm(nframes, block, bps, seed=99 + level)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm_h).to(dev)
    first = 16000                                     # (frame numbers 16000 ...: two- and three-byte UTF-8 numbers in one batch)
    settings = flac_amd.make_settings(2, bps, rate, level)

    def encode(max_batch):
        eng = flac_amd.FrameEngine(settings, device=0, max_batch_frames=max_batch)
        try:
            cap = eng.max_output_bytes(max_batch)
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_fb = torch.empty(max_batch, dtype=torch.int32, device=dev)
            d_total = torch.zeros(1, dtype=torch.int64, device=dev)
            parts, fbs, kernels = [], [], set()
            for f0 in range(0, nframes, max_batch):
                nf = min(max_batch, nframes - f0)
                eng.encode_device(d_pcm.data_ptr() + f0 * block * 2 * 4, nf, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(), first_frame_number=first + f0)
                torch.cuda.synchronize()
                parts.append(d_out[: int(d_total.item())].cpu())
                fbs.append(d_fb[:nf].cpu())
                kernels |= set(eng.last_batch_kernels())
            return torch.cat(parts), torch.cat(fbs), kernels
        finally:
            eng.close()

    small, fb_small, _ = encode(16384)
    big, fb_big, kernels = encode(nframes)
    assert torch.equal(fb_big, fb_small), "frame lengths differ, first at frame %d" % int((fb_big != fb_small).nonzero()[0])
    assert big.numel() == small.numel() and torch.equal(big, small), sorted(kernels)
    assert int(fb_big.to(torch.int64).sum()) == big.numel()
    # the kernels the bench's step runs are the ones that ran here
    if level >= 5 and bps == 16:
        assert {"prep3_kernel", "autoc3_kernel", "evalg_kernel", "pack2_kernel"} <= kernels, sorted(kernels)
        if level == 8:
            assert "autoc3_kernel<SETS>" in kernels, sorted(kernels)      # (three window-job sets: a wavefront per set from 1.5 rounds of the chip's slots up)
