"""-m gpu: 17..24-bit stereo at the presets without an LPC search (-0, -1, -2) in their 1152-sample blocks: ff_kernel<MS, -1, WIDE>
(round 6: the one-kernel frame -- a wavefront per frame, a lane owns 18 samples of both channels from the load to the last Rice
code -- with a register per sample and channel, 64-bit difference sums for the 25-bit side channel, the Rice search in 32-bit node
arithmetic while no partition sum reaches 2^31 and on 64-bit sums beyond, codes of up to 31 bits, frames of up to 7.5 KB) against the
oracle, byte for byte, and the kernel record.  The reference serves these widths by the same code with its wide sums:
stream_encoder.c:4098-4108 (fixed predictor by sample width), fixed.c:301, stream_encoder.c:4701-5075 (partition search),
:4814-4817 (32- or 64-bit partition sums), :4786-4791 (RICE2 when a parameter reaches 15)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


def _encode(s, pcm, **kw):
    import flac_amd
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=kw.get("batch", 16))
    try:
        data, fb = eng.encode(pcm)
        ks = eng.last_batch_kernels()
    finally:
        eng.close()
    return data, fb, ks


def _cases(n, bps, seed):
    rng = np.random.default_rng(seed)
    fs = 1 << (bps - 1)
    yield "music", signals.music(n, 2, bps, seed=seed)
    yield "full-scale noise", rng.integers(-fs, fs, size=(n, 2)).astype(np.int32)
    alt = np.where(np.arange(n) % 2 == 0, fs - 1, -fs)
    yield "full-scale alternation in anti-phase (the widest side channel)", np.stack([alt, -alt - 1], axis=1).astype(np.int32)
    yield "full-scale alternation in phase", np.stack([alt, alt], axis=1).astype(np.int32)
    yield "wasted bits", (signals.music(n, 2, bps, seed=seed + 1) >> 5) << 5
    yield "tiny", rng.integers(-3, 4, size=(n, 2)).astype(np.int32)
    yield "constant left, noise right", np.stack([np.full(n, 12345 % fs), rng.integers(-fs, fs, size=n)], axis=1).astype(np.int32)
    yield "both constant", np.stack([np.full(n, -7), np.full(n, 3)], axis=1).astype(np.int32)
    yield "digital silence", np.zeros((n, 2), dtype=np.int32)
    # noise of every loudness: a lane's |residual| sum walks through 2^23 (the 16-bit kernel's limit), 2^25 (where this one switches
    # to 64-bit sums) and on to full scale -- each frame of the signal another amplitude
    amp = np.repeat((fs * 2.0 ** (-np.linspace(0, 14, n // 1152 + 1))).astype(np.int64), 1152)[:n]
    yield "noise, a loudness per frame", (rng.integers(-1 << 30, 1 << 30, size=(n, 2)) * amp[:, None] >> 30).astype(np.int32)
    yield "impulses", np.where(rng.random((n, 2)) < 0.002, rng.integers(-fs, fs, size=(n, 2)), 0).astype(np.int32)
    t = np.arange(n)
    yield "a loud low tone and a quiet right channel", np.stack([(0.98 * fs * np.sin(t * 0.01)).astype(np.int64), rng.integers(-40, 40, size=n)], axis=1).astype(np.int32)


@pytest.mark.parametrize("bps", [17, 20, 24])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_wide_stereo_in_one_kernel(bps, level, monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    n = 1152 * 40 + 333
    s = flac_amd.make_settings(2, bps, 96000, level, streamable_subset=0)
    for name, pcm in _cases(n, bps, 7 * bps + level):
        data, fb, ks = _encode(s, pcm)
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, bps, level)
        assert "ff_kernel" in ks and "prep2_kernel<DECIDE>" not in ks, (name, ks)         # (the short last block has the general kernels)


@pytest.mark.parametrize("extra", [dict(min_partition_order=0, max_partition_order=6), dict(min_partition_order=1, max_partition_order=3), dict(min_partition_order=2, max_partition_order=2),
                                   dict(max_partition_order=0), dict(mid_side=1, loose_mid_side=1), dict(limit_min_bitrate=1),
                                   dict(disable=(1, 0, 0)), dict(disable=(0, 0, 1)), dict(disable=(0, 1, 0)), dict(disable=(0, 1, 1))])
def test_wide_stereo_in_one_kernel_with_other_settings(extra, monkeypatch):
    """partition order ranges other than the presets' 0..3 (down to 18-sample partitions), the loose mid/side search, the subframe
    types switched off, the minimum bit rate's constant rule -- at 24 and 21 bits"""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    n = 1152 * 24 + 77
    for bps, level in ((24, 2), (21, 0), (24, 1)):
        s = flac_amd.make_settings(2, bps, 48000, level, streamable_subset=0, **extra)
        for name, pcm in _cases(n, bps, 1000 + bps):
            data, fb, ks = _encode(s, pcm)
            o = oracle_encode_settings(pcm, s)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, bps, level, extra)
            # (partitions of 18 samples are not pack2_kernel's, and what is not pack2_kernel's is not ff_kernel's: the same line as at 16 bits)
            assert "ff_kernel" in ks or extra.get("max_partition_order", 3) > 3, (name, ks, extra)


def test_the_wide_flavour_can_be_switched_off_and_16_bits_do_not_take_it(monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    s = flac_amd.make_settings(2, 24, 96000, 0, streamable_subset=0)
    pcm = signals.music(1152 * 9, 2, 24, seed=5)
    o = oracle_encode_settings(pcm, s)
    monkeypatch.setenv("FLACGPU_NO_WIDE_FF", "1")
    data, fb, ks = _encode(s, pcm)
    assert data == o["data"] and "ff_kernel" not in ks and "prep2_kernel<DECIDE>" in ks, ks
    monkeypatch.delenv("FLACGPU_NO_WIDE_FF")
    data, fb, ks = _encode(s, pcm)
    assert data == o["data"] and "ff_kernel" in ks, ks


def test_a_large_batch_of_wide_frames_and_the_verify_pass(monkeypatch):
    """20 000 frames of 24-bit music at -2 in one launch: every CRC-16 rechecked on the host (the walk over the stream), 300 frames
    against the oracle; the same batch with the device verify pass on -- which takes pack2_kernel's frames, whose run starts its
    decoder wants, as at 16 bits -- must be the same bytes and verify"""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    nf = 20000
    s = flac_amd.make_settings(2, 24, 96000, 2, streamable_subset=0)
    base = signals.music(1152 * 500, 2, 24, seed=11)
    pcm = np.concatenate([np.roll(base, 1152 * 37 * i, axis=0) for i in range(nf // 500)])
    out = []
    for verify in (False, True):
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nf)
        try:
            if verify:
                eng.set_verify(True)
            data, fb = eng.encode(pcm)
            ks = eng.last_batch_kernels()
            if verify:
                v = eng.last_verify_result()
                assert v.status == 0, (v.status, v.frame_number, v.sample)
        finally:
            eng.close()
        assert ("ff_kernel" in ks) == (not verify), ks
        assert len(fb) == nf
        out.append((data, fb))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    data, fb = out[0]
    o = oracle_encode_settings(pcm[:1152 * 300], s)
    nb = int(np.sum(o["frame_bytes"]))
    assert np.array_equal(fb[:300], o["frame_bytes"]) and data[:nb] == o["data"]
    # every frame of the batch: sync code, CRC-16
    crc_tab = np.zeros(256, dtype=np.uint16)
    for i in range(256):
        c = i << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
        crc_tab[i] = c
    buf = np.frombuffer(data, dtype=np.uint8)
    off = 0
    for f in range(0, nf, 97):
        off = int(np.sum(fb[:f]))
        fr = buf[off:off + int(fb[f])]
        assert fr[0] == 0xff and (fr[1] & 0xfe) == 0xf8
        c = 0
        for b in fr[:-2]:
            c = ((c << 8) & 0xffff) ^ int(crc_tab[(c >> 8) ^ int(b)])
        assert c == (int(fr[-2]) << 8 | int(fr[-1])), f


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FLACGPU_FF_SEEDS", "60"))))
def test_adversarial_signals_on_the_one_kernel_frame(seed, monkeypatch):
    """The adversarial signals of tests/test_adversarial_cpu.py (resonances, exact polynomials, impulses, full-scale patterns, random
    walks, related channels, wasted bits -- the oracle is held to the reference on them there) as stereo in 1152-sample blocks at the
    presets without an LPC search, 8..24 bits, WITHOUT the verify pass (which would take the frames away from ff_kernel: the seeded
    sweep and tests/test_adversarial_gpu.py run with it on), random subframe switches and partition order ranges."""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    from test_adversarial_cpu import adversarial_signal
    monkeypatch.setenv("FLACGPU_POISON", "1")
    rng = np.random.default_rng(424200 + seed)
    for sub in range(6):
        bps = int(rng.choice([8, 12, 16, 16, 17, 18, 20, 21, 23, 24, 24, 24]))
        level = int(rng.integers(0, 3))
        n = 1152 * int(rng.integers(1, 9)) + (int(rng.integers(0, 1152)) if rng.random() < 0.5 else 0)
        kw = dict(streamable_subset=0, limit_min_bitrate=int(rng.random() < 0.3))
        if rng.random() < 0.3:
            kw["disable"] = tuple(int(rng.random() < 0.4) for _ in range(3))
        if rng.random() < 0.4:
            hi = int(rng.integers(0, 4))
            kw["max_partition_order"], kw["min_partition_order"] = hi, int(rng.integers(0, hi + 1))
        pcm = adversarial_signal(rng, n, 2, bps)
        s = flac_amd.make_settings(2, bps, int(rng.choice([44100, 48000, 96000, 12345])), level, **kw)
        batch = int(rng.integers(1, 12))
        data, fb, ks = _encode(s, pcm, batch=batch)
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (seed, sub, bps, level, n, kw)
        # (the record is the last batch's: it has whole blocks in it when the batch holds more than the short last block)
        assert "ff_kernel" in ks or "disable" in kw or (n % 1152 and (n // 1152) % batch == 0), (seed, sub, bps, level, n, kw, batch, ks)
