"""CPU: the host-side verify decoder (flac_amd/csrc/host/verify.c, FLAC__stream_encoder_set_verify) decodes frames of the
oracle / the reference back to the input, locates an audio mismatch, and rejects damaged frames."""
import ctypes as C

import numpy as np
import pytest

import signals
from flac_amd import engine
from oracle import pyoracle as po


class VerifyResult(C.Structure):
    _fields_ = [("status", C.c_int), ("frame_number", C.c_uint32), ("channel", C.c_uint32), ("sample", C.c_uint32),
                ("absolute_sample", C.c_uint64), ("expected", C.c_int32), ("got", C.c_int32)]


def _verify(pcm, bps, rate, level, frames, fb, first_frame=0, threads=3, **kw):
    lib = engine.load_host()
    lib.flacgpu_host_verify_batch.restype = C.c_int
    lib.flacgpu_host_verify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_uint32, C.c_uint32, C.POINTER(VerifyResult)]
    s = engine.make_settings(pcm.shape[1], bps, rate, level, **kw)
    width = (bps + 7) // 8
    raw = np.zeros((pcm.size, width), dtype=np.uint8)
    flat = pcm.reshape(-1).astype(np.int64)
    for k in range(width):
        raw[:, k] = (flat >> (8 * k)) & 0xff
    N = s.blocksize
    n = pcm.shape[0]
    nf = (n + N - 1) // N
    tail = n - (nf - 1) * N
    tail = 0 if tail == N else tail
    fbuf = np.frombuffer(frames, dtype=np.uint8).copy()
    fb = np.ascontiguousarray(fb, dtype=np.uint32)
    res = VerifyResult()
    st = lib.flacgpu_host_verify_batch(C.byref(s), fbuf.ctypes.data, fb.ctypes.data, nf, tail, first_frame, raw.ctypes.data, width, threads, C.byref(res))
    return st, res


@pytest.mark.parametrize("level", [0, 1, 2, 3, 5, 8])
def test_oracle_frames_verify(level):
    for fam, ch, bps, rate in (("music", 2, 16, 44100), ("mixed", 2, 16, 44100), ("wasted", 2, 16, 44100), ("square", 2, 16, 44100),
                               ("music", 1, 16, 44100), ("music", 2, 24, 96000), ("music", 3, 16, 48000), ("quiet", 2, 16, 44100)):
        pcm = signals.FAMILIES[fam](4096 * 3 + 321, ch, bps)
        o = po.oracle_encode(pcm, bps, rate, level, first_frame=126)      # frame numbers across the 1- / 2-byte UTF-8 boundary
        st, res = _verify(pcm, bps, rate, level, o["data"], o["frame_bytes"], first_frame=126)
        assert st == 0, (fam, ch, bps, level, res.status, res.frame_number)


def test_mismatch_is_located_and_damage_is_rejected():
    pcm = signals.music(4096 * 4 + 100, 2, 16, seed=4)
    o = po.oracle_encode(pcm, 16, 44100, 8)
    # the frames are fine, the "input" differs in one sample: mismatch at exactly that place
    bad = pcm.copy()
    bad[2 * 4096 + 17, 1] += 3
    st, res = _verify(bad, 16, 44100, 8, o["data"], o["frame_bytes"])
    assert st == 1 and (res.frame_number, res.channel, res.sample, res.absolute_sample) == (2, 1, 17, 2 * 4096 + 17)
    assert (res.expected, res.got) == (int(bad[2 * 4096 + 17, 1]), int(pcm[2 * 4096 + 17, 1]))
    # a flipped bit in the third frame: its CRC-16 no longer matches
    off = np.concatenate([[0], np.cumsum(o["frame_bytes"].astype(np.int64))])
    data = bytearray(o["data"])
    data[int(off[3]) + 40] ^= 0x10
    st, res = _verify(pcm, 16, 44100, 8, bytes(data), o["frame_bytes"])
    assert st == 2 and res.frame_number == 3


def test_reference_frames_verify_with_escape_free_rice2():
    if not po.have_ref():
        pytest.skip("no oracle/_ref")
    pcm = signals.white(4096 * 2 + 5, 2, 24)                # 24-bit noise: Rice2 parameters
    r = po.ref_encode(pcm, 24, 96000, 5)
    frames = r["data"][r["header_bytes"]:]
    st, res = _verify(pcm, 24, 96000, 5, frames, r["frame_bytes"])
    assert st == 0


@pytest.mark.parametrize("bps", [25, 28, 32])
def test_wide_sample_frames_verify(bps):
    """more than 24 bits per sample; at 32 the side channel is read back as 33-bit samples"""
    for fam, ch in (("music", 2), ("white", 2), ("square", 2), ("wasted", 2), ("music", 1)):
        for level in (0, 2, 5, 8):
            pcm = signals.FAMILIES[fam](4096 * 2 + 321, ch, bps)
            o = po.oracle_encode(pcm, bps, 96000, level)
            st, res = _verify(pcm, bps, 96000, level, o["data"], o["frame_bytes"], streamable_subset=0)
            assert st == 0, (fam, ch, bps, level, res.status, res.frame_number)
    for pattern in range(1, 8):
        pcm = signals.fsd(4096 + 100, 2, bps, pattern)
        o = po.oracle_encode(pcm, bps, 96000, 5)
        st, res = _verify(pcm, bps, 96000, 5, o["data"], o["frame_bytes"], streamable_subset=0)
        assert st == 0, (pattern, bps, res.status, res.frame_number)


def test_crc_recheck_of_a_run_of_frames_finds_the_first_damaged_one():
    """flacgpu_host_check_frame_crcs (the corpus job's and bench.py's every-frame CRC-16 recheck) on oracle frames"""
    import ctypes as C
    import numpy as np
    import signals
    from flac_amd import engine
    from oracle import pyoracle as po
    host = engine.load_host()
    host.flacgpu_host_check_frame_crcs.restype = C.c_int64
    host.flacgpu_host_check_frame_crcs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    pcm = signals.music(4096 * 37 + 900, 2, 16, seed=3)
    enc = po.oracle_encode(pcm, 16, 44100, 5)
    data = np.frombuffer(enc["data"], dtype=np.uint8).copy()
    fb = np.ascontiguousarray(enc["frame_bytes"], dtype=np.uint32)
    for threads in (1, 3, 8):
        assert host.flacgpu_host_check_frame_crcs(data.ctypes.data, fb.ctypes.data, fb.size, threads) == -1
    offs = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
    for f in (0, 17, 36, fb.size - 1):
        d = data.copy()
        d[offs[f] + int(fb[f]) // 2] ^= 0x10
        d[offs[min(f + 5, fb.size - 1)] + 3] ^= 0x01          # a later one as well: the FIRST is reported
        for threads in (1, 4):
            assert host.flacgpu_host_check_frame_crcs(d.ctypes.data, fb.ctypes.data, fb.size, threads) == f
