"""-m gpu: streams of more than 16 bits whose channels hold 16-bit pairs after the wasted bits are out (ChanPrep::fmt = 1) -- 16-bit audio
in a 24-bit container, 12-bit audio in 20 bits, a quiet side channel under loud left / right -- on evalw_kernel, which takes planes of
pairs since round 6 (they went to the general evaluation body, 2.9x the time of the same audio as a 16-bit stream), against the oracle.
The reference shifts the wasted bits out per subframe and evaluates on the narrow signal (stream_encoder.c:3833-3870, 4098-4108)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _check(pcm, ch, bps, level, want_evalg, what, **kw):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    kw.setdefault("streamable_subset", 0)
    s = flac_amd.make_settings(ch, bps, 48000, level, **kw)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=64)
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (what, sorted(ks))
    if want_evalg:
        assert "evalw_kernel" in ks, (what, sorted(ks))


@pytest.mark.parametrize("level", [3, 5, 8])
@pytest.mark.parametrize("ch", [1, 2, 6])
def test_narrow_audio_in_a_wide_container(level, ch, monkeypatch):
    monkeypatch.setenv("FLACGPU_POISON", "1")
    n = 4096 * 9                       # whole blocks: the kernel record is the nominal frames'
    a16 = signals.music(n, ch, 16, seed=level + ch)
    a12 = signals.music(n, ch, 12, seed=level)
    a24 = signals.music(n, ch, 24, seed=3)
    rng = np.random.default_rng(level * 10 + ch)
    for name, bps, pcm in (("16 in 24", 24, a16 << 8), ("12 in 20", 20, a12 << 8), ("16 in 17", 17, a16 << 1), ("16 in 32", 32, a16.astype(np.int64) << 16),
                           ("noise16 in 24", 24, rng.integers(-32768, 32768, size=(n, ch)) << 8), ("full-scale square 16 in 24", 24, signals.fullscale_square(n, ch, 16) << 8)):
        _check(np.ascontiguousarray(pcm, dtype=np.int64).astype(np.int32), ch, bps, level, bps <= 24, (name, level, ch))
    # channels of one stream apart: the first narrow, the others wide
    if ch > 1:
        mix = a24.copy()
        mix[:, 0] = a16[:, 0] << 8
        _check(mix, ch, 24, level, True, ("first channel 16 in 24", level, ch))
    # a quiet side channel under loud 24-bit left / right (its plane holds 16-bit pairs although the stream is wide)
    if ch == 2:
        side = rng.integers(-2000, 2000, size=n)
        q = a24.copy()
        q[:, 1] = q[:, 0] - side
        _check(q, 2, 24, level, True, ("quiet side", level))
    # the short last block rides along
    _check((a16[: 4096 * 3 + 1000] << 8).astype(np.int32), ch, 24, level, True, ("16 in 24 with a short last block", level, ch))


@pytest.mark.parametrize("blocksize", [1024, 4608, 8192, 1152])
def test_narrow_audio_in_a_wide_container_other_block_sizes(blocksize, monkeypatch):
    """(1152: evalw_kernel does not serve 18-sample runs -- the stream keeps the general kernel, and must still come out right)"""
    monkeypatch.setenv("FLACGPU_POISON", "1")
    a16 = signals.music(blocksize * 7 + 55, 2, 16, seed=blocksize)
    _check((a16 << 8).astype(np.int32), 2, 24, 8, blocksize != 1152, ("16 in 24", blocksize), blocksize=blocksize, streamable_subset=0)


def test_narrow_audio_in_a_wide_container_many_frames(monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    nframes = 600
    pcm = (signals.music(nframes * 4096, 2, 16, seed=77) << 8).astype(np.int32)
    s = flac_amd.make_settings(2, 24, 96000, 8)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nframes)
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    assert "evalw_kernel" in ks, sorted(ks)
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"]
