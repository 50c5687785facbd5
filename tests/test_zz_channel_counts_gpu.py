"""-m gpu, last file of the suite on purpose: every channel count 1..8 through the HIP path against the oracle.

The seeded sweeps draw their channel counts from 1, 2, 3, 6, 8 and the layout tests name 1, 2, 3, 4, 6, 8: five and seven channels --
the two counts whose rounds in prep4_kernel are uneven (3 + 2, 4 + 3 channels per round; flacgpu_prep.hip) -- were in no test.  The
oracle is held to the reference on all eight counts in tests/test_oracle_vs_ref.py::test_every_channel_count.  Written after the
round's GPU budget was spent: this file had not run on hardware when it was committed ("Open")."""
import numpy as np
import pytest

import signals
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("forced", [False, True], ids=["default selection", "streaming autocorrelation forced"])
@pytest.mark.parametrize("ch", range(1, 9))
def test_every_channel_count_gpu_vs_oracle(ch, forced, monkeypatch):
    import flac_amd
    if forced:
        monkeypatch.setenv("FLACGPU_AUTOC2", "1")
        monkeypatch.setenv("FLACGPU_AUTOC3", "1")
    for bps, rate in ((16, 48000), (24, 96000)):
        base = signals.music(4096 * 5 + 333, ch, bps, seed=10 * ch + bps)
        flat = base.copy()
        flat[:, :] = 5
        flat[:, -1] = base[:, -1] if ch > 1 else 5
        one_const = base.copy()
        one_const[:, ch // 2] = -7
        for level, pcm, kw, okw in ((8, base, {}, {}), (5, base, {}, {}), (0, base, dict(blocksize=4096), dict(blocksize=4096)),
                                    (8, flat, dict(limit_min_bitrate=1), dict(limit_min_bitrate=1)),
                                    (5, one_const, dict(limit_min_bitrate=1), dict(limit_min_bitrate=1)),
                                    (8, np.zeros_like(base), dict(limit_min_bitrate=1), dict(limit_min_bitrate=1))):
            eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, rate, level, **kw), device=0, max_batch_frames=8)
            try:
                eng.set_verify(True)
                data, fb = eng.encode(pcm)
                v = eng.last_verify_result()
                kernels = eng.last_batch_kernels()
            finally:
                eng.close()
            assert v.status == 0, ("verify", ch, bps, level, kw, v.status, v.frame_number, v.channel, v.sample)
            o = po.oracle_encode(pcm, bps, rate, level, **okw)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (ch, bps, level, kw, sorted(kernels))


@pytest.mark.parametrize("seed", range(20))
def test_adversarial_signals_with_any_channel_count_gpu_vs_oracle(seed, monkeypatch):
    """the adversarial cases of tests/test_adversarial_cpu.py with the channel count drawn from 1..8 (there: oracle == reference on
    40 000 such cases); like the test above, not yet run on hardware when committed"""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    from test_adversarial_cpu import adversarial_case
    monkeypatch.setenv("FLACGPU_POISON", "1")
    for sub in range(8):
        pcm, ch, bps, rate, kw, s = adversarial_case(seed * 8 + sub, True)
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
        try:
            eng.set_verify(True)
            data, fb = eng.encode(pcm)
            v = eng.last_verify_result()
            assert v.status == 0, ("verify", v.status, v.frame_number, v.channel, v.sample, seed, sub, ch, bps, rate, kw)
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (seed, sub, ch, bps, rate, kw)
