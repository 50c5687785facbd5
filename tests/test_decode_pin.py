"""The GPU verify decoder's logic, pinned on the CPU: flac_amd/csrc/flacgpu_decode.h (the code a lane of the verify
kernels runs) compiled for the host (oracle/libdecodepin.so) and driven like the kernels drive it, against
 * frames of the oracle over the configuration space: everything must decode back to its input;
 * the host frame decoder (host/verify.c, flacgpu_host_verify_batch): same verdict and the same located mismatch when
   the input differs from what was encoded, same "does not decode" when a frame is damaged."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import signals
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_SO = os.path.join(ROOT, "oracle", "libdecodepin.so")


class PinResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("frame_number", C.c_uint32), ("channel", C.c_uint32), ("sample", C.c_uint32),
                ("absolute_sample", C.c_uint64), ("expected", C.c_int32), ("got", C.c_int32)]


class HostResult(C.Structure):
    _fields_ = [("status", C.c_int), ("frame_number", C.c_uint32), ("channel", C.c_uint32), ("sample", C.c_uint32),
                ("absolute_sample", C.c_uint64), ("expected", C.c_int32), ("got", C.c_int32)]


def _pin():
    if not os.path.exists(PIN_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    lib = C.CDLL(PIN_SO)
    lib.decodepin_verify_batch.restype = C.c_int
    lib.decodepin_verify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                           C.c_void_p, C.c_uint32, C.POINTER(PinResult)]
    return lib


def pin_verify(data, fb, pcm, bps, blocksize, first=0, maxord=32):
    lib = _pin()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    nfr = len(fb)
    tail = n - (nfr - 1) * blocksize
    tail = 0 if tail == blocksize else tail
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    fbs = np.ascontiguousarray(fb, dtype=np.uint32)
    full = np.zeros((nfr * blocksize, ch), dtype=np.int32)
    full[:n] = pcm
    r = PinResult()
    st = lib.decodepin_verify_batch(buf.ctypes.data, fbs.ctypes.data, nfr, ch, bps, blocksize, tail, first, full.ctypes.data, maxord, C.byref(r))
    return st, r


def host_verify(data, fb, pcm, bps, blocksize, first=0):
    from flac_amd import engine
    host = engine.load_host()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    s = engine.make_settings(ch, bps, 44100, 5, blocksize=blocksize, streamable_subset=0)
    width = (bps + 7) // 8
    raw = np.zeros((n, ch, 4), dtype=np.uint8)
    raw[:] = pcm.astype("<i4").view(np.uint8).reshape(n, ch, 4)
    raw = np.ascontiguousarray(raw[:, :, :width])
    nfr = len(fb)
    tail = n - (nfr - 1) * blocksize
    tail = 0 if tail == blocksize else tail
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    fbs = np.ascontiguousarray(fb, dtype=np.uint32)
    r = HostResult()
    host.flacgpu_host_verify_batch.restype = C.c_int
    host.flacgpu_host_verify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    st = host.flacgpu_host_verify_batch(C.byref(s), buf.ctypes.data, fbs.ctypes.data, nfr, tail, first, raw.ctypes.data, width, 2, C.byref(r))
    return st, r


CASES = [
    # (channels, bps, level, blocksize, nsamples, family, kwargs)
    (2, 16, 8, 4096, 4096 * 5 + 1000, "music", {}),
    (2, 16, 5, 4096, 4096 * 3, "white", {}),
    (2, 16, 0, 1152, 1152 * 7 + 13, "music", {}),
    (1, 16, 8, 4096, 4096 * 3 + 17, "music", {}),
    (2, 24, 8, 4096, 4096 * 3 + 999, "music", {}),
    (2, 8, 5, 4096, 4096 * 3, "music", {}),
    (2, 32, 8, 4096, 4096 * 2 + 100, "music", {}),
    (2, 32, 5, 4096, 4096 * 2, "white", {}),
    (6, 16, 5, 4096, 4096 * 2 + 5, "music", {}),
    (8, 24, 8, 2048, 2048 * 3, "music", {}),
    (2, 16, 8, 256, 256 * 9 + 31, "music", {}),
    (2, 16, 5, 4096, 4096 * 3, "silence", {}),
    (2, 16, 5, 4096, 4096 * 3, "constant", {}),
    (2, 16, 5, 4096, 4096 * 3, "wasted", {}),
    (2, 16, 8, 4096, 4096 * 3, "square", {}),
    (2, 20, 8, 4608, 4608 * 2 + 77, "music", {}),
    (2, 16, 8, 16384, 16384 * 2, "music", {"max_lpc_order": 32}),
    (2, 12, 3, 576, 576 * 5, "music", {}),
    (2, 16, 8, 4096, 4096 * 6 + 3, "mixed", {}),
    (2, 24, 8, 4096, 4096 * 3, "quiet", {}),
    (2, 16, 8, 4096, 4096 * 2, "music", {"exhaustive": 1}),
    (3, 16, 5, 1024, 1024 * 4 + 1, "white", {}),
]


def _signal(family, n, ch, bps, seed):
    if family == "silence":
        return signals.silence(n, ch, bps)
    if family == "constant":
        return signals.constant(n, ch, bps)
    if family == "square":
        return signals.fullscale_square(n, ch, bps)
    if family == "mixed":
        return signals.mixed(n, ch, bps, seed=seed)
    return getattr(signals, family)(n, ch, bps, seed=seed)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_oracle_frames_decode_back_to_their_input(case):
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, 11)
    enc = po.oracle_encode(pcm, bps, 44100, level, first_frame=123456, blocksize=bs, **kw)
    st, r = pin_verify(enc["data"], enc["frame_bytes"], pcm, bps, bs, first=123456)
    assert st == 0, (st, r.frame_number, r.channel, r.sample, r.expected, r.got)
    hst, _ = host_verify(enc["data"], enc["frame_bytes"], pcm, bps, bs, first=123456)
    assert hst == 0


@pytest.mark.parametrize("case", CASES[:11], ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_mismatch_is_located_like_the_host_decoder_locates_it(case):
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, 12)
    enc = po.oracle_encode(pcm, bps, 44100, level, first_frame=7, blocksize=bs, **kw)
    rng = np.random.default_rng(5)
    for trial in range(6):
        bad = pcm.copy()
        i, c = int(rng.integers(0, n)), int(rng.integers(0, ch))
        bad[i, c] ^= 1 << int(rng.integers(0, bps - 1))
        if trial % 2:                                     # a second difference later in the stream: the FIRST is reported
            j = min(n - 1, i + int(rng.integers(1, 3 * bs)))
            bad[j, (c + 1) % ch] ^= 1
        st, r = pin_verify(enc["data"], enc["frame_bytes"], bad, bps, bs, first=7)
        hst, h = host_verify(enc["data"], enc["frame_bytes"], bad, bps, bs, first=7)
        assert st == hst == 1
        assert (r.frame_number, r.channel, r.sample, r.absolute_sample, r.expected, r.got) == (h.frame_number, h.channel, h.sample, h.absolute_sample, h.expected, h.got)
        assert r.frame_number == 7 + i // bs and r.sample == i % bs and r.channel == c


@pytest.mark.parametrize("case", CASES[:6], ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_damaged_frames_do_not_decode(case):
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, 13)
    enc = po.oracle_encode(pcm, bps, 44100, level, blocksize=bs, **kw)
    fb = enc["frame_bytes"]
    offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
    rng = np.random.default_rng(9)
    for trial in range(12):
        d = bytearray(enc["data"])
        f = int(rng.integers(0, len(fb)))
        pos = int(offs[f]) + int(rng.integers(0, fb[f]))
        d[pos] ^= 1 << int(rng.integers(0, 8))
        st, r = pin_verify(bytes(d), fb, pcm, bps, bs)
        hst, h = host_verify(bytes(d), fb, pcm, bps, bs)
        assert st == hst == 2 and r.frame_number == h.frame_number == f


def test_damage_that_keeps_the_crc_is_still_caught():
    """a frame whose CRC-16 was recomputed after the damage: the body itself must fail to decode or to match"""
    lib = po.load_oracle()
    pcm = signals.music(4096 * 4, 2, 16, seed=21)
    enc = po.oracle_encode(pcm, 16, 44100, 8)
    fb = enc["frame_bytes"]
    offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
    rng = np.random.default_rng(3)
    for trial in range(40):
        d = np.frombuffer(enc["data"], dtype=np.uint8).copy()
        f = int(rng.integers(0, len(fb)))
        pos = int(offs[f]) + int(rng.integers(6, fb[f] - 2))
        d[pos] ^= 1 << int(rng.integers(0, 8))
        body = d[offs[f]:offs[f + 1] - 2]
        crc = lib.fo_crc16(body.ctypes.data, body.size)
        d[offs[f + 1] - 2], d[offs[f + 1] - 1] = crc >> 8, crc & 0xff
        st, r = pin_verify(d.tobytes(), fb, pcm, 16, 4096)
        hst, h = host_verify(d.tobytes(), fb, pcm, 16, 4096)
        assert st in (1, 2) and st == hst and r.frame_number == h.frame_number == f
        if st == 1:
            assert (r.channel, r.sample, r.expected, r.got) == (h.channel, h.sample, h.expected, h.got)


class _Bits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for k in range(n - 1, -1, -1):
            self.bits.append((v >> k) & 1)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def test_escaped_partitions_and_rice2_decode():
    """No encoder here emits escape codes (the reference dropped do_escape_coding), so a frame is built by hand: FIXED order 2,
    partition order 2 with an escaped partition of 9-bit raw residuals, a Rice partition, an escaped partition of width 0
    (all zero) and a Rice partition with parameter 0; then the same as RICE2 with a 5-bit parameter of 17."""
    lib = po.load_oracle()
    N, order = 64, 2
    rng = np.random.default_rng(2)
    for method in (0, 1):
        res = np.zeros(N, dtype=np.int64)
        res[order:16] = rng.integers(-200, 200, 16 - order)
        res[16:32] = rng.integers(-9, 9, 16)
        res[32:48] = 0
        res[48:64] = rng.integers(-2, 2, 16)
        if method:
            res[16:32] = rng.integers(-(1 << 19), 1 << 19, 16)
        x = np.zeros(N, dtype=np.int64)
        x[0], x[1] = 100, 90
        for i in range(order, N):
            x[i] = res[i] + 2 * x[i - 1] - x[i - 2]
        bps = 16 if not method else 32
        if np.abs(x).max() >= 1 << (bps - 1):
            x[:] = 0
        w = _Bits()
        w.put(0x3ffe, 14); w.put(0, 1); w.put(0, 1)
        w.put(6, 4); w.put(9, 4); w.put(0, 4); w.put(4 if bps == 16 else 7, 3); w.put(0, 1)
        w.put(5, 8)                                    # frame number 5
        w.put(N - 1, 8)
        hdr = w.bytes()
        w.put(lib.fo_crc8(hdr, len(hdr)), 8)
        w.put(0, 1); w.put(8 + order, 6); w.put(0, 1)
        for i in range(order):
            w.put(int(x[i]) & ((1 << bps) - 1), bps)
        plen, esc = (5, 31) if method else (4, 15)
        w.put(method, 2); w.put(2, 4)

        def rice(v, k):
            u = (int(v) << 1) ^ (int(v) >> 63)
            u &= (1 << 64) - 1
            w.put(0, u >> k); w.put(1, 1)
            if k:
                w.put(u & ((1 << k) - 1), k)
        w.put(esc, plen); w.put(9, 5)
        for i in range(order, 16):
            w.put(int(res[i]) & 0x1ff, 9)
        k1 = 17 if method else 3
        w.put(k1, plen)
        for i in range(16, 32):
            rice(res[i], k1)
        w.put(esc, plen); w.put(0, 5)
        w.put(0, plen)
        for i in range(48, 64):
            rice(res[i], 0)
        body = w.bytes()
        crc = lib.fo_crc16(body, len(body))
        frame = body + bytes([crc >> 8, crc & 0xff])
        pcm = x.astype(np.int32).reshape(N, 1)
        fb = np.array([len(frame)], dtype=np.uint32)
        st, r = pin_verify(frame, fb, pcm, bps, N, first=5)
        assert st == 0, (method, st, r.sample, r.expected, r.got)
        hst, _ = host_verify(frame, fb, pcm, bps, N, first=5)
        assert hst == 0
        bad = pcm.copy()
        bad[40, 0] += 1                              # inside the all-zero escaped partition
        st, r = pin_verify(frame, fb, bad, bps, N, first=5)
        assert st == 1 and r.sample == 40 and r.got == int(x[40]) and r.expected == int(x[40]) + 1


# ---- the hinted pass (flacgpu_decode_hinted.h): a thread per 16-sample run, hints from the pack kernel, hints NOT trusted --------
HRUNS = 256


def _hinted_lib():
    lib = _pin()
    lib.decodepin_make_hints.restype = C.c_int
    lib.decodepin_make_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.decodepin_verify_hinted.restype = C.c_int
    lib.decodepin_verify_hinted.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_void_p]
    return lib


def _frames_args(data, fb, pcm, blocksize):
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    nfr = len(fb)
    tail = n - (nfr - 1) * blocksize
    tail = 0 if tail == blocksize else tail
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    fbs = np.ascontiguousarray(fb, dtype=np.uint32)
    full = np.zeros((nfr * blocksize, ch), dtype=np.int32)
    full[:n] = pcm
    return buf, fbs, full, nfr, ch, tail


def make_hints(data, fb, pcm, bps, blocksize, first=0):
    lib = _hinted_lib()
    buf, fbs, full, nfr, ch, tail = _frames_args(data, fb, pcm, blocksize)
    hints = np.zeros((nfr, ch, HRUNS), dtype=np.uint32)
    covered = np.zeros(nfr, dtype=np.uint8)
    lib.decodepin_make_hints(buf.ctypes.data, fbs.ctypes.data, nfr, ch, bps, blocksize, tail, first, hints.ctypes.data, covered.ctypes.data)
    return hints, covered.astype(bool)


def hinted_verify(data, fb, pcm, bps, blocksize, hints, first=0, maxord=16):
    lib = _hinted_lib()
    buf, fbs, full, nfr, ch, tail = _frames_args(data, fb, pcm, blocksize)
    hints = np.ascontiguousarray(hints, dtype=np.uint32)
    suspect = np.zeros(nfr, dtype=np.uint8)
    lib.decodepin_verify_hinted(buf.ctypes.data, fbs.ctypes.data, nfr, ch, bps, blocksize, tail, first, full.ctypes.data, hints.ctypes.data, maxord, suspect.ctypes.data)
    return suspect.astype(bool)


def _per_frame_sequential(data, fb, pcm, bps, bs, first=0):
    """the sequential decoder's verdict for every frame on its own: True = fine"""
    offs = np.concatenate([[0], np.cumsum(np.asarray(fb, dtype=np.int64))])
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    ok = []
    for f in range(len(fb)):
        part = pcm[f * bs:(f + 1) * bs]
        st, _ = pin_verify(bytes(data[offs[f]:offs[f + 1]]), fb[f:f + 1], part, bps, len(part) if len(part) < bs else bs, first=first + f)
        ok.append(st == 0)
    return np.array(ok)


# configurations the hinted pass covers: blocks of up to 4096 samples in whole 16-sample runs, orders up to 16, at most 32 bits
HINTED_CASES = [c for c in CASES if c[3] <= 4096 and c[3] % 16 == 0 and "max_lpc_order" not in c[6]]


@pytest.mark.parametrize("case", HINTED_CASES, ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_hinted_pass_accepts_the_oracles_frames_with_honest_hints(case):
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, 31)
    enc = po.oracle_encode(pcm, bps, 44100, level, first_frame=77, blocksize=bs, **kw)
    hints, covered = make_hints(enc["data"], enc["frame_bytes"], pcm, bps, bs, first=77)
    suspect = hinted_verify(enc["data"], enc["frame_bytes"], pcm, bps, bs, hints, first=77)
    nfull = n // bs
    wide = bps == 32 and ch == 2                      # a 33-bit side channel can occur: such subframes are the sequential decoder's
    for f in range(len(covered)):
        if f < nfull and covered[f] and not wide:
            assert not suspect[f], f
    if not wide:
        assert covered[:nfull].all()                   # every full frame of these configurations is within the pass


@pytest.mark.parametrize("case", HINTED_CASES[:8], ids=lambda c: "%dch-%db-l%d-bs%d-%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_hinted_pass_flags_every_changed_sample(case):
    ch, bps, level, bs, n, family, kw = case
    pcm = _signal(family, n, ch, bps, 32)
    enc = po.oracle_encode(pcm, bps, 44100, level, blocksize=bs, **kw)
    hints, covered = make_hints(enc["data"], enc["frame_bytes"], pcm, bps, bs)
    rng = np.random.default_rng(8)
    for trial in range(12):
        bad = pcm.copy()
        i, c = int(rng.integers(0, (n // bs) * bs)), int(rng.integers(0, ch))
        bad[i, c] ^= 1 << int(rng.integers(0, bps - 1))
        suspect = hinted_verify(enc["data"], enc["frame_bytes"], bad, bps, bs, hints)
        assert suspect[i // bs]
        others = np.delete(np.arange(len(suspect)), i // bs)
        assert not suspect[others][covered[others] & (others < n // bs)].any() or (bps == 32 and ch == 2)


def test_hinted_pass_is_sound_under_damage_and_arbitrary_hints():
    """whatever is done to the frame bytes (CRC aside: its own kernel checks that) or to the hints: a frame the hinted pass
    accepts is a frame the sequential decoder accepts"""
    rng = np.random.default_rng(2024)
    pcm = signals.music(4096 * 6, 2, 16, seed=41)
    for level in (8, 5, 2):
        bs = 4096 if level >= 3 else 1152
        pcm_l = pcm[:(len(pcm) // bs) * bs]
        enc = po.oracle_encode(pcm_l, 16, 44100, level, blocksize=bs)
        fb = enc["frame_bytes"]
        offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
        hints, covered = make_hints(enc["data"], fb, pcm_l, 16, bs)
        if bs % 16 == 0:
            assert covered.all()
        accepted_damaged = 0
        for trial in range(150):
            d = np.frombuffer(enc["data"], dtype=np.uint8).copy()
            h = hints.copy()
            f = int(rng.integers(0, len(fb)))
            kind = trial % 5
            if kind in (0, 1):                                    # a flipped bit anywhere in the body
                pos = int(offs[f]) + int(rng.integers(0, fb[f] - 2))
                d[pos] ^= 1 << int(rng.integers(0, 8))
            if kind in (1, 2):                                    # a hint moved by a few bits
                c, t = int(rng.integers(0, 2)), int(rng.integers(0, bs // 16))
                h[f, c, t] = max(0, int(h[f, c, t]) + int(rng.integers(-40, 41)))
            if kind == 3:                                         # hints of another frame
                h[f] = hints[(f + 1) % len(fb)]
            if kind == 4:                                         # random hints
                h[f] = rng.integers(0, 8 * int(fb[f]), size=h[f].shape, dtype=np.uint32)
            suspect = hinted_verify(d.tobytes(), fb, pcm_l, 16, bs, h)
            seq_ok = _per_frame_sequential(d.tobytes(), fb, pcm_l, 16, bs)
            # (the sequential verdict here ignores the CRC-16, as the hinted pass does: compare bodies)
            for g in range(len(fb)):
                if not suspect[g]:
                    body_ok = seq_ok[g]
                    if not body_ok:
                        # the only way the sequential pin rejects what the hinted pass accepts is the CRC footer it also checks
                        dd = d[offs[g]:offs[g + 1]]
                        lib = po.load_oracle()
                        crc = lib.fo_crc16(dd[:-2].ctypes.data, dd.size - 2)
                        assert (int(dd[-2]) << 8 | int(dd[-1])) != crc, (level, trial, g)
                        accepted_damaged += 1
            if kind in (2, 3, 4) and bs % 16 == 0:
                assert suspect[f] or np.array_equal(h[f], hints[f])   # wrong hints never verify a frame
        # a flipped bit that the hinted pass accepts must be one that changes nothing but the CRC's validity: there is none in the body
        assert accepted_damaged == 0


def test_the_two_frame_header_parsers_agree():
    """hinted_frame_header (five words at once) against decode_frame_header (the bit reader): every frame number length, block
    size and sample rate code the encoder can emit, at every alignment; and the same verdict on every single-bit flip of a header
    and on random bytes"""
    lib = _pin()
    lib.decodepin_header_both.restype = C.c_int
    lib.decodepin_header_both.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    rng = np.random.default_rng(3)
    seen_ok = 0
    for bs, rate, bps, ch in ((4096, 44100, 16, 2), (1152, 48000, 24, 2), (4000, 44100, 16, 1), (200, 12345, 8, 2), (65535, 96000, 32, 3), (256, 655350, 12, 6),
                              (4608, 8000, 20, 2), (16, 11025, 16, 8)):
        n = bs * 2
        pcm = signals.music(n, ch, bps, seed=bs)
        for first in (0, 127, 128, 2047, 2048, 65535, 65536, 2 ** 21 - 1, 2 ** 21, 2 ** 26 - 1, 2 ** 26, 2 ** 31 - 2):
            enc = po.oracle_encode(pcm[:bs], bps, rate, 2, first_frame=first, blocksize=bs)
            frame = np.frombuffer(enc["data"], dtype=np.uint8)
            for lead in range(4):
                buf = np.concatenate([np.full(lead, 0x5A, np.uint8), frame, np.zeros(32, np.uint8)])
                # (the buffer itself is 16-byte aligned by numpy: `lead` sets the frame's alignment)
                r = lib.decodepin_header_both(buf.ctypes.data, len(frame), lead, ch, bps, bs, bs, first)
                assert r == 0, (bs, rate, bps, ch, first, lead, hex(r))
                seen_ok += 1
            # every single-bit flip of the first 16 bytes, and a wrong expectation: the two parsers fail or pass together
            for bit in range(16 * 8):
                d = frame.copy()
                d[bit // 8] ^= 0x80 >> (bit % 8)
                buf = np.concatenate([np.full(1, 0x5A, np.uint8), d, np.zeros(32, np.uint8)])
                r = lib.decodepin_header_both(buf.ctypes.data, len(d), 1, ch, bps, bs, bs, first)
                assert (r & 0xff) == ((r >> 8) & 0xff) and not (r >> 16), (bs, first, bit, hex(r))
            r = lib.decodepin_header_both(buf.ctypes.data, len(frame), 1, ch, bps, bs, bs, first + 1)
            assert (r & 0xff) == ((r >> 8) & 0xff) == 2
    for trial in range(3000):
        d = rng.integers(0, 256, size=64, dtype=np.uint8)
        d[0], d[1] = 0xff, 0xf8                                   # let most of them past the sync code
        lead = int(rng.integers(0, 4))
        buf = np.concatenate([np.full(lead, 0x11, np.uint8), d, np.zeros(32, np.uint8)])
        r = lib.decodepin_header_both(buf.ctypes.data, 64, lead, 2, 16, 4096, 4096, int(rng.integers(0, 200)))
        assert (r & 0xff) == ((r >> 8) & 0xff) and not (r >> 16), (trial, hex(r))
    assert seen_ok > 300
