"""-m gpu: 17..24-bit input at the presets without an LPC search (-0, -1, -2): the deciding prep kernel's wide flavour
(prep2_kernel<WIDE, ., DECIDE>, round 6: chunk sums and leaf sums in 64 bits, the partition-order search on them where a leaf leaves
the 32-bit node arithmetic) against the oracle -- and the kernel record: these shapes used to put EVERY channel on eval_list_kernel's
list (24-bit -0: 37 G samples/s against 154 G at 16 bits, slower than -5).  Reference: stream_encoder.c:4098-4108 (the wide
fixed-predictor routines by sample width), :4701-5075 (partition search), :4814-4817 (32- or 64-bit partition sums)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _signals(n, ch, bps, seed):
    rng = np.random.default_rng(seed)
    fs = 1 << (bps - 1)
    yield "music", signals.music(n, ch, bps, seed=seed)
    yield "full-scale noise", rng.integers(-fs, fs, size=(n, ch)).astype(np.int32)
    alt = np.where(np.arange(n) % 2 == 0, fs - 1, -fs)
    yield "full-scale alternation, channels in anti-phase", np.stack([alt if c % 2 == 0 else -alt - 1 for c in range(ch)], axis=1).astype(np.int32)
    q = (signals.music(n, ch, bps, seed=seed + 1) >> 5) << 5
    yield "wasted bits", q
    tiny = rng.integers(-3, 4, size=(n, ch)).astype(np.int32)
    yield "tiny", tiny


@pytest.mark.parametrize("bps", [17, 20, 24])
@pytest.mark.parametrize("ch", [1, 2, 3])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_wide_input_at_the_fixed_only_presets(bps, ch, level, monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    for blocksize in (1152, 2304, 4608, 576):
        n = blocksize * 5 + 321
        s = flac_amd.make_settings(ch, bps, 96000, level, blocksize=blocksize, streamable_subset=0)
        for name, pcm in _signals(n, ch, bps, 31 * bps + ch):
            eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
            try:
                data, fb = eng.encode(pcm)
                ks = eng.last_batch_kernels()
            finally:
                eng.close()
            o = oracle_encode_settings(pcm, s)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, bps, ch, level, blocksize)
    # the preset's own block size: the record must show the deciding kernel and nothing of the general evaluation for the whole blocks
    s = flac_amd.make_settings(ch, bps, 96000, level, streamable_subset=0)
    pcm = signals.music(1152 * 6, ch, bps, seed=3)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    # (stereo in the presets' 1152-sample blocks is ff_kernel<., ., WIDE>'s since the end of round 6: tests/test_wide_ff_gpu.py)
    assert ("ff_kernel" in ks if ch == 2 else "prep2_kernel<DECIDE>" in ks) and "eval_kernel" not in ks, ks
