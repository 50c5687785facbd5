"""-m gpu: prep3_kernel<., NW, CH> (round 6) -- the stereo mid/side prep kernel at the block sizes that are NW x 64 x CH samples: 1024,
2048, 8192 with 16-sample chunks, 1152, 2304, 4608 with 18-sample chunks (4096 = <., 4, 16> as ever) -- against the oracle, with the
kernel record.  These sizes ran prep2_kernel at the LPC presets (2.7x the time per sample: `-8 -b 1152` 1.39x of `-8`);
the reference sets any of them with FLAC__stream_encoder_set_blocksize (stream_encoder.c:1716, test/test_streams.sh:172-219)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

SIZES = [1024, 2048, 8192, 1152, 2304, 4608]


def _check(pcm, s, what, want_prep3=True, frames=16):
    from oracle_from_settings import oracle_encode_settings
    import flac_amd
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=frames)
    try:
        # whole blocks only first: the kernel record of a batch without the short last block
        nb = (len(pcm) // s.blocksize) * s.blocksize
        data, fb = eng.encode(pcm[:nb])
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    o = oracle_encode_settings(pcm[:nb], s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], what
    if want_prep3:
        assert "prep3_kernel" in ks and "prep2_kernel" not in ks, (what, ks)
    # and the stream with its short last block (the general kernels take that one)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=frames)
    try:
        data, fb = eng.encode(pcm)
    finally:
        eng.close()
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (what, "with the last block")


@pytest.mark.parametrize("blocksize", SIZES)
@pytest.mark.parametrize("level", [4, 5, 8])
@pytest.mark.parametrize("bps", [16, 24])
def test_prep3_at_other_block_sizes(blocksize, level, bps, monkeypatch):
    import flac_amd
    monkeypatch.setenv("FLACGPU_POISON", "1")
    s = flac_amd.make_settings(2, bps, 44100, level, blocksize=blocksize, streamable_subset=0)
    rng = np.random.default_rng(blocksize * 10 + level)
    n = blocksize * 7 + 77
    fs = 1 << (bps - 1)
    const = np.full((n, 2), 1234, dtype=np.int32)
    const[:, 1] = -77
    left_const = signals.music(n, 2, bps, seed=3).copy()
    left_const[:, 0] = 5
    for name, pcm in (("music", signals.music(n, 2, bps, seed=level)), ("noise", rng.integers(-fs, fs, size=(n, 2)).astype(np.int32)),
                      ("quiet", rng.integers(-3, 4, size=(n, 2)).astype(np.int32)), ("wasted", (signals.music(n, 2, bps, seed=9) >> 4) << 4),
                      ("constant", const), ("left constant", left_const)):
        _check(pcm, s, (name, blocksize, level, bps))


@pytest.mark.parametrize("blocksize", SIZES)
def test_prep3_other_sizes_sums_at_the_32_bit_edge(blocksize, monkeypatch):
    """the part sums of fourth differences around 2^32 (ADVICE r05's case, at these parts' lengths: 64 x 16 or 64 x 18 samples), the
    side channel's 18 differences per lane in 64 bits from 18 bits up, the full mid/side search and the loose one"""
    import flac_amd
    monkeypatch.setenv("FLACGPU_POISON", "1")
    q = 64 * (18 if blocksize % 1152 == 0 else 16)
    n = blocksize * 3
    sign = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int64)
    for bps in (16, 17, 18, 20, 24):
        fs = 1 << (bps - 1)
        wrap = (1 << 32) / (q * 16.0)
        for mult in (0.51, 0.97, 1.03, 2.02, 1e9):
            a = int(min(fs - 1, round(wrap * mult)))
            if a < 1:
                continue
            pcm = np.stack([sign * a, -sign * (a - 1)], axis=1).astype(np.int32)
            for level, loose in ((5, 0), (8, 0), (5, 1)):
                s = flac_amd.make_settings(2, bps, 48000, level, blocksize=blocksize, streamable_subset=0, mid_side=1, loose_mid_side=loose)
                _check(pcm, s, ("edge", blocksize, bps, a, level, loose))


@pytest.mark.parametrize("blocksize,level", [(1152, 8), (2304, 5), (8192, 8), (1024, 5)])
def test_prep3_other_sizes_many_frames(blocksize, level, monkeypatch):
    """thousands of workgroups in one launch, every frame against the oracle"""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    s = flac_amd.make_settings(2, 16, 44100, level, blocksize=blocksize, streamable_subset=0)
    nframes = 1500000 // blocksize
    pcm = signals.music(nframes * blocksize, 2, 16, seed=blocksize)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nframes)
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    assert "prep3_kernel" in ks and "prep2_kernel" not in ks, ks
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"]


# ---- prep4_kernel<., NW, CH>: independent channels (mono, stereo without a mid/side search, 3..8 channels) at the same block sizes ----
def _check4(pcm, s, what, frames=16):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    nb = (len(pcm) // s.blocksize) * s.blocksize
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=frames)
    try:
        data, fb = eng.encode(pcm[:nb])
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    o = oracle_encode_settings(pcm[:nb], s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], what
    assert "prep4_kernel" in ks and "prep2_kernel" not in ks, (what, ks)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=frames)
    try:
        data, fb = eng.encode(pcm)
    finally:
        eng.close()
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (what, "with the last block")


@pytest.mark.parametrize("blocksize", SIZES)
@pytest.mark.parametrize("ch,bps,level", [(1, 16, 8), (1, 24, 5), (2, 16, 3), (3, 16, 5), (5, 20, 8), (6, 24, 5), (7, 16, 3), (8, 16, 8), (8, 24, 3)])
def test_prep4_at_other_block_sizes(blocksize, ch, bps, level, monkeypatch):
    """rounds of 1..4 channels (two at most with eight wavefronts), uneven last rounds (5 = 3 + 2, 7 = 4 + 3), every sample width's sums"""
    import flac_amd
    monkeypatch.setenv("FLACGPU_POISON", "1")
    kw = dict(mid_side=0) if ch == 2 else {}
    s = flac_amd.make_settings(ch, bps, 48000, level, blocksize=blocksize, streamable_subset=0, **kw)
    rng = np.random.default_rng(blocksize * 10 + ch)
    n = blocksize * 5 + 33
    fs = 1 << (bps - 1)
    const = np.tile(np.arange(ch, dtype=np.int32) * 100 - 50, (n, 1))
    lastc = signals.music(n, ch, bps, seed=4).copy()
    if ch > 1:
        lastc[:, :-1] = 7                                            # every channel but the last constant
    for name, pcm in (("music", signals.music(n, ch, bps, seed=level)), ("noise", rng.integers(-fs, fs, size=(n, ch)).astype(np.int32)),
                      ("wasted", (signals.music(n, ch, bps, seed=9) >> 3) << 3), ("constant", const), ("all but the last constant", lastc)):
        _check4(pcm, s, (name, blocksize, ch, bps, level))
    # limit_min_bitrate (stream_encoder.c:3874-3879): the last channel of an all-constant frame is not CONSTANT -- the rule reads the other channels' records
    s2 = flac_amd.make_settings(ch, bps, 48000, level, blocksize=blocksize, streamable_subset=0, limit_min_bitrate=1, **kw)
    _check4(const, s2, ("limit_min_bitrate, all constant", blocksize, ch, bps, level))
    _check4(lastc, s2, ("limit_min_bitrate, all but the last constant", blocksize, ch, bps, level))


@pytest.mark.parametrize("blocksize", SIZES)
def test_prep4_other_sizes_sums_at_the_32_bit_edge(blocksize, monkeypatch):
    import flac_amd
    monkeypatch.setenv("FLACGPU_POISON", "1")
    q = 64 * (18 if blocksize % 1152 == 0 else 16)
    n = blocksize * 3
    sign = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int64)
    for bps in (17, 18, 19, 20, 24):
        fs = 1 << (bps - 1)
        wrap = (1 << 32) / (q * 16.0)
        for mult in (0.51, 0.97, 1.03, 2.02, 1e9):
            a = int(min(fs - 1, round(wrap * mult)))
            if a < 1:
                continue
            for ch in (1, 3):
                pcm = np.stack([sign * a if c % 2 == 0 else -sign * (a - c) for c in range(ch)], axis=1).astype(np.int32)
                s = flac_amd.make_settings(ch, bps, 48000, 5, blocksize=blocksize, streamable_subset=0)
                _check4(pcm, s, ("edge", blocksize, bps, a, ch))


@pytest.mark.parametrize("blocksize,ch,level", [(1152, 1, 8), (4608, 6, 5), (8192, 2, 8), (2048, 1, 5)])
def test_prep4_other_sizes_many_frames(blocksize, ch, level, monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    kw = dict(mid_side=0) if ch == 2 else {}
    s = flac_amd.make_settings(ch, 16, 48000, level, blocksize=blocksize, streamable_subset=0, **kw)
    nframes = 1200000 // (blocksize * ch)
    pcm = signals.music(nframes * blocksize, ch, 16, seed=blocksize)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nframes)
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    assert "prep4_kernel" in ks and "prep2_kernel" not in ks, ks
    o = oracle_encode_settings(pcm, s)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"]
