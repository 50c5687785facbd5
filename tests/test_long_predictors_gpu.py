"""-m gpu: predictors of 13..32 taps (-l 13 .. -l 32, FLAC__MAX_LPC_ORDER, format.h:128; the reference's own matrix runs -l 16 and
-l 32 -e -p, test/test_streams.sh:182-208) on the wavefront-per-channel evaluation kernel (round 6: evalg_kernel<32>, a chain of up to
17 v_dot2_i32_i16 in two asm statements, windows of three pieces reaching in front of a lane's run) against the oracle, with the
kernel record.  These orders ran on the general evaluation kernel: -8 -l 32 took ten times -8 per sample
(profiles/r06_n_order_rate.txt).

The reference forms such predictors' sums in 64 bits on 16-bit input (bps + precision + ilog2(order) > 32, lpc.c:942-976); the
kernel takes a candidate when the bound on its 32-bit sum holds, and must hand the others -- full-scale signals with large
coefficients -- to the general kernel: the loud and the resonant signals below are there for that path."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _signals(n, ch, bps, seed):
    rng = np.random.default_rng(seed)
    fs = 1 << (bps - 1)
    t = np.arange(n)
    # a full-scale resonance: predictors with large coefficients, sums near the 32-bit bound
    res = np.clip(np.rint((fs - 1) * 0.98 * np.sin(2 * np.pi * t * 0.0113) * (0.6 + 0.4 * np.sin(2 * np.pi * t * 0.00071))), -fs, fs - 1).astype(np.int32)
    square = np.where((t // 37) % 2 == 0, fs - 1, -fs).astype(np.int32)
    out = [("music", signals.music(n, ch, bps, seed=seed)), ("noise", rng.integers(-fs, fs, size=(n, ch)).astype(np.int32)),
           ("quiet", rng.integers(-3, 4, size=(n, ch)).astype(np.int32)), ("wasted", (signals.music(n, ch, bps, seed=seed + 1) >> 3) << 3),
           ("resonance", np.stack([res if c % 2 == 0 else np.roll(res, 5) for c in range(ch)], axis=1)),
           ("square", np.stack([square if c % 2 == 0 else -square - 1 for c in range(ch)], axis=1))]
    return [(k, np.ascontiguousarray(v, dtype=np.int32)) for k, v in out]


@pytest.mark.parametrize("order", [13, 16, 17, 20, 24, 31, 32])
@pytest.mark.parametrize("level", [5, 8])
def test_long_predictors_on_the_fast_evaluation(order, level, monkeypatch):
    _run(order, level, 2, 16, 4096, {}, monkeypatch, expect_evalg=True)


@pytest.mark.parametrize("ch,bps,blocksize,kw", [(1, 16, 4096, {}), (2, 16, 8192, {}), (2, 16, 3072, {}), (2, 12, 4096, {}), (2, 16, 4096, dict(exhaustive=1)),
                                                 (2, 16, 4096, dict(prec_search=1)), (6, 16, 4096, {}), (2, 16, 2048, {}), (2, 24, 4096, {}), (2, 16, 4608, {})],
                         ids=["mono", "b8192", "b3072", "12-bit", "-e", "-p", "5.1", "b2048 (runs too short)", "24-bit (not this kernel's)", "b4608 (half pieces)"])
def test_long_predictors_other_shapes(ch, bps, blocksize, kw, monkeypatch):
    for order in (16, 32):
        _run(order, 8, ch, bps, blocksize, kw, monkeypatch, expect_evalg=None)


def test_forced_off_is_the_general_kernel(monkeypatch):
    monkeypatch.setenv("FLACGPU_NO_EVALG32", "1")
    _run(32, 8, 2, 16, 4096, {}, monkeypatch, expect_evalg=False)


def _run(order, level, ch, bps, blocksize, kw, monkeypatch, expect_evalg):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    s = flac_amd.make_settings(ch, bps, 44100, level, blocksize=blocksize, max_lpc_order=order, streamable_subset=0, **kw)
    n = blocksize * 7 + 333
    for name, pcm in _signals(n, ch, bps, order + level):
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=16)
        try:
            data, fb = eng.encode(pcm)
            ks = eng.last_batch_kernels()
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, order, level, ch, bps, blocksize, sorted(ks))
        if expect_evalg is True:
            assert "evalg_kernel" in ks, (name, order, level, sorted(ks))
        elif expect_evalg is False:
            assert "evalg_kernel" not in ks, (name, order, level, sorted(ks))
