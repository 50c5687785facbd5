"""The self check on the device: flacgpu_verify_batch_device / flacgpu_set_verify decode the frames of a
batch on the GPU (one lane per frame, flac_amd/csrc/flacgpu_verify.hip) and compare them with the input.
 * every configuration family: what the engine encoded decodes back to its input;
 * an input that differs from what was encoded is located exactly like the host frame decoder (host/verify.c) locates it,
   which in turn is how the reference reports it (stream_encoder.c:3000-3018, 5186-5230);
 * damaged frames are rejected at the same frame."""
import ctypes as C

import numpy as np
import pytest

import flac_amd
import signals
from test_decode_pin import CASES, _signal, host_verify

pytestmark = pytest.mark.gpu


def _vr(t):
    v = flac_amd.VerifyResult.from_buffer_copy(t.cpu().numpy().tobytes())
    return v


def _setup(case, seed, first=0, level_kw=None):
    import torch
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, seed)
    kw2 = dict(kw)
    s = flac_amd.make_settings(ch, bps, 44100, level, blocksize=bs, streamable_subset=0, **{k: v for k, v in kw2.items() if k in ("max_lpc_order", "exhaustive")})
    nfr = (n + bs - 1) // bs
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nfr)
    data, fb = eng.encode(pcm, first_frame_number=first)
    return eng, pcm, data, fb, bs


def _device_verify(eng, data, fb, pcm, bs, first=0):
    import torch
    n, ch = pcm.shape
    nfr = len(fb)
    tail = n - (nfr - 1) * bs
    tail = 0 if tail == bs else tail
    full = np.zeros((nfr * bs, ch), dtype=np.int32)
    full[:n] = pcm
    d_frames = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    d_fb = torch.from_numpy(np.ascontiguousarray(fb, dtype=np.uint32).view(np.int32)).cuda()
    d_pcm = torch.from_numpy(full).cuda()
    d_res = torch.zeros(32, dtype=torch.uint8, device="cuda")
    eng.verify_device(d_frames.data_ptr(), d_fb.data_ptr(), nfr, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=first, tail=tail,
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return _vr(d_res)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_engine_frames_verify_on_the_device(case):
    eng, pcm, data, fb, bs = _setup(case, 31, first=70000)
    try:
        v = _device_verify(eng, data, fb, pcm, bs, first=70000)
        assert v.status == 0, (v.status, v.frame_number, v.channel, v.sample, v.expected, v.got)
        # frames in front of an unaligned address: the same batch behind 1..3 stray bytes
        import torch
        n, ch = pcm.shape
        nfr = len(fb)
        tail = n - (nfr - 1) * bs
        tail = 0 if tail == bs else tail
        full = np.zeros((nfr * bs, ch), dtype=np.int32)
        full[:n] = pcm
        d_pcm = torch.from_numpy(full).cuda()
        d_fb = torch.from_numpy(np.ascontiguousarray(fb, dtype=np.uint32).view(np.int32)).cuda()
        for shift in (1, 3):
            buf = torch.from_numpy(np.concatenate([np.full(shift, 0xAA, np.uint8), np.frombuffer(data, dtype=np.uint8)])).cuda()
            d_res = torch.zeros(32, dtype=torch.uint8, device="cuda")
            eng.verify_device(buf.data_ptr() + shift, d_fb.data_ptr(), nfr, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=70000, tail=tail)
            torch.cuda.synchronize()
            assert _vr(d_res).status == 0
    finally:
        eng.close()


@pytest.mark.parametrize("case", CASES[:11], ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_mismatch_and_damage_are_reported_like_the_host_decoder_reports_them(case):
    ch, bps, level, bs, n, family, kw = case
    eng, pcm, data, fb, bs = _setup(case, 32, first=9)
    try:
        rng = np.random.default_rng(17)
        for trial in range(4):
            bad = pcm.copy()
            i, c = int(rng.integers(0, n)), int(rng.integers(0, ch))
            bad[i, c] ^= 1 << int(rng.integers(0, bps - 1))
            if trial % 2:
                j = min(n - 1, i + int(rng.integers(1, 3 * bs)))
                bad[j, (c + 1) % ch] ^= 1
            v = _device_verify(eng, data, fb, bad, bs, first=9)
            hst, h = host_verify(data, fb, bad, bps, bs, first=9)
            assert v.status == hst == 1
            assert (v.frame_number, v.channel, v.sample, v.absolute_sample, v.expected, v.got) == (h.frame_number, h.channel, h.sample, h.absolute_sample, h.expected, h.got)
        offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
        for trial in range(6):
            d = bytearray(data)
            f = int(rng.integers(0, len(fb)))
            pos = int(offs[f]) + int(rng.integers(0, fb[f]))
            d[pos] ^= 1 << int(rng.integers(0, 8))
            v = _device_verify(eng, bytes(d), fb, pcm, bs, first=9)
            hst, h = host_verify(bytes(d), fb, pcm, bps, bs, first=9)
            assert v.status == hst == 2 and v.frame_number == h.frame_number == 9 + f
    finally:
        eng.close()


def test_set_verify_checks_every_batch_of_the_host_entry_points():
    pcm = signals.music(4096 * 300 + 77, 2, 16, seed=5)
    s = flac_amd.make_settings(2, 16, 44100, 8)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=128)
    try:
        eng.set_verify(True)
        data, fb = eng.encode(pcm)                  # three batches, the last one with a short block
        v = eng.last_verify_result()
        assert v.status == 0
        from oracle import pyoracle as po
        assert data == po.oracle_encode(pcm, 16, 44100, 8)["data"]
        raw = pcm[:4096 * 100].astype("<i2").tobytes()
        out, fb2 = eng.encode_raw(raw, flac_amd.raw_format(16))
        assert eng.last_verify_result().status == 0
    finally:
        eng.close()


def test_large_batch_verifies_and_one_damaged_frame_in_16384_is_found():
    import torch
    nframes = 4096
    pcm = signals.music(4096 * 512, 2, 16, seed=8)
    pcm = np.concatenate([pcm] * (nframes // 512), axis=0)
    s = flac_amd.make_settings(2, 16, 44100, 8)
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nframes)
    try:
        d_pcm = torch.from_numpy(pcm).cuda()
        cap = eng.max_output_bytes(nframes)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_fb = torch.empty(nframes, dtype=torch.int32, device="cuda")
        d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
        d_res = torch.zeros(32, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        eng.encode_device(d_pcm.data_ptr(), nframes, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(), first_frame_number=1000, stream=st)
        eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nframes, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=1000, stream=st)
        torch.cuda.synchronize()
        assert _vr(d_res).status == 0
        offs = torch.cumsum(d_fb.to(torch.int64), 0)
        f = 2777
        d_out[int(offs[f - 1].item()) + 100] ^= 0x40
        d_out[int(offs[3000].item()) + 50] ^= 0x01                  # a later one as well
        eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nframes, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=1000, stream=st)
        torch.cuda.synchronize()
        v = _vr(d_res)
        assert v.status == 2 and v.frame_number == 1000 + f
    finally:
        eng.close()


# ---- the hinted pass (flacgpu_decode_hinted.h, verify_hinted_kernel): a thread per 16-sample run, from the pack kernel's hints -----
def _encode_on_device(eng, pcm, nfr, first):
    import torch
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    cap = eng.max_output_bytes(nfr)
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_fb = torch.zeros(nfr, dtype=torch.int32, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    eng.encode_device(d_pcm.data_ptr(), nfr, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(), first_frame_number=first)
    torch.cuda.synchronize()
    return d_pcm, d_out, d_fb, int(d_total.item())


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_hinted_pass_vouches_for_the_frames_it_covers(case):
    """verify right after the encode, on the encode's own output: the frames of configurations the pass covers (blocks of up to 4096
    samples in whole 16-sample runs, orders up to 16, no 33-bit side channel) never reach the sequential decoder"""
    import torch
    ch, bps, level, bs, n, family, kw = case
    nfr = n // bs                                              # full blocks only: the short last block is the general pack kernel's
    pcm = _signal(family, n, ch, bps, 33)[:nfr * bs]
    s = flac_amd.make_settings(ch, bps, 44100, level, blocksize=bs, streamable_subset=0, **{k: v for k, v in kw.items() if k in ("max_lpc_order", "exhaustive")})
    eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=nfr)
    try:
        eng.set_verify(True)                                   # (allocates the verify buffers, the hints among them)
        d_pcm, d_out, d_fb, total = _encode_on_device(eng, pcm, nfr, 500)
        d_res = torch.zeros(32, dtype=torch.uint8, device="cuda")
        eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nfr, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=500)
        torch.cuda.synchronize()
        assert _vr(d_res).status == 0
        hinted = eng.verify_hinted_frames()
        covered = bs <= 4096 and bs % 16 == 0 and not (bps == 32 and ch == 2) and kw.get("max_lpc_order", 0) <= 16
        if bs == 4096 and covered and ch <= 2:
            assert hinted == nfr, (hinted, nfr)                 # (smaller blocks may pick partitions shorter than a run: those frames go the long way)
        elif not covered:
            assert hinted == 0
    finally:
        eng.close()


def test_hinted_pass_and_sequential_decoder_give_one_verdict(monkeypatch):
    """FLACGPU_VERIFY_FORCE_HINTS=1: the hints of the last encode are used whatever frames are handed in -- here copies with a
    changed input sample, a flipped bit (the CRC re-made, so that the body itself must give it away), hints that belong to other
    frames: the verdict and its location are the host decoder's, and the frames the pass vouched for are all the others"""
    import torch
    from oracle import pyoracle as po
    monkeypatch.setenv("FLACGPU_VERIFY_FORCE_HINTS", "1")
    lib = po.load_oracle()
    rng = np.random.default_rng(77)
    for level, bps in ((8, 16), (5, 16), (8, 24)):
        nfr, bs = 24, 4096
        pcm = signals.music(bs * nfr, 2, bps, seed=50 + level)
        eng = flac_amd.FrameEngine(flac_amd.make_settings(2, bps, 44100, level), device=0, max_batch_frames=nfr)
        try:
            eng.set_verify(True)
            d_pcm, d_out, d_fb, total = _encode_on_device(eng, pcm, nfr, 9)
            data = d_out[:total].cpu().numpy().tobytes()
            fb = d_fb.cpu().numpy().astype(np.uint32)
            offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
            v = _device_verify(eng, data, fb, pcm, bs, first=9)          # a copy of the frames: forced hints
            assert v.status == 0 and eng.verify_hinted_frames() == nfr
            for trial in range(8):
                bad = pcm.copy()
                i, c = int(rng.integers(0, bs * nfr)), int(rng.integers(0, 2))
                bad[i, c] ^= 1 << int(rng.integers(0, bps - 1))
                v = _device_verify(eng, data, fb, bad, bs, first=9)
                hst, h = host_verify(data, fb, bad, bps, bs, first=9)
                assert v.status == hst == 1
                assert (v.frame_number, v.channel, v.sample, v.expected, v.got) == (h.frame_number, h.channel, h.sample, h.expected, h.got)
                assert eng.verify_hinted_frames() == nfr - 1
            for trial in range(12):
                d = np.frombuffer(data, dtype=np.uint8).copy()
                f = int(rng.integers(0, nfr))
                pos = int(offs[f]) + int(rng.integers(6, fb[f] - 2))
                d[pos] ^= 1 << int(rng.integers(0, 8))
                body = d[offs[f]:offs[f + 1] - 2]
                crc = lib.fo_crc16(body.ctypes.data, body.size)
                d[offs[f + 1] - 2], d[offs[f + 1] - 1] = crc >> 8, crc & 0xff
                v = _device_verify(eng, d.tobytes(), fb, pcm, bs, first=9)
                hst, h = host_verify(d.tobytes(), fb, pcm, bps, bs, first=9)
                assert v.status == hst and v.status in (1, 2) and v.frame_number == h.frame_number == 9 + f
                if hst == 1:
                    assert (v.channel, v.sample, v.expected, v.got) == (h.channel, h.sample, h.expected, h.got)
                assert eng.verify_hinted_frames() == nfr - 1
            # another batch's frames under this batch's hints: nothing is vouched for wrongly, the batch still verifies (sequentially)
            pcm2 = signals.music(bs * nfr, 2, bps, seed=90 + level)
            eng2 = flac_amd.FrameEngine(flac_amd.make_settings(2, bps, 44100, level), device=0, max_batch_frames=nfr)
            data2, fb2 = eng2.encode(pcm2, first_frame_number=9)
            eng2.close()
            v = _device_verify(eng, data2, fb2, pcm2, bs, first=9)
            assert v.status == 0
            assert eng.verify_hinted_frames() < nfr            # (a frame of silence could chain by accident; music does not)
        finally:
            eng.close()


def test_in_place_damage_after_the_encode_is_found_through_the_hinted_path():
    import torch
    nfr, bs = 512, 4096
    pcm = np.concatenate([signals.music(bs * 128, 2, 16, seed=61)] * 4, axis=0)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8), device=0, max_batch_frames=nfr)
    try:
        eng.set_verify(True)
        d_pcm, d_out, d_fb, total = _encode_on_device(eng, pcm, nfr, 0)
        d_res = torch.zeros(32, dtype=torch.uint8, device="cuda")
        offs = torch.cumsum(d_fb.to(torch.int64), 0)
        d_out[int(offs[300].item()) + 200] ^= 0x10               # frame 301
        d_pcm[bs * 100 + 7, 1] += 1                              # frame 100: the first problem in stream order
        eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nfr, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=0)
        torch.cuda.synchronize()
        v = _vr(d_res)
        assert (v.status, v.frame_number, v.channel, v.sample) == (1, 100, 1, 7)
        assert eng.verify_hinted_frames() == nfr - 2
    finally:
        eng.close()
