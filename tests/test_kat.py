"""Known-answer tests taken from the reference's own unit tests:
CRC definitions (src/test_libFLAC/crc.c:60-95), MD5 digests (src/test_libFLAC/md5.c:36-221),
UTF-8 frame-number coding (src/test_libFLAC/bitwriter.c:280-400 / bitwriter.c:832)."""
import ctypes as C
import hashlib

import numpy as np

from oracle import pyoracle as po
from flac_amd import engine


def _crc8_ref(data):
    crc = 0
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = ((crc << 1) ^ (0x07 if crc >> 7 else 0)) & 0xFF
    return crc


def _crc16_ref(data):
    crc = 0
    for b in data:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ (0x8005 if crc >> 15 else 0)) & 0xFFFF
    return crc


def test_crc_matches_reference_test_definition():
    lib = po.load_oracle()
    data = bytearray(2048)          # the reference test's reproducible pseudo-random fill (crc.c:46-47)
    for i in range(1, len(data)):
        c = data[i - 1] ^ (i % 256)
        for _ in range(8):
            c = ((c << 1) ^ (0x07 if c >> 7 else 0)) & 0xFF
        data[i] = c
    buf = (C.c_uint8 * len(data)).from_buffer(data)
    for n in (0, 1, 2, 7, 8, 63, 64, 1000, 2048):
        assert lib.fo_crc8(buf, n) == _crc8_ref(data[:n])
        assert lib.fo_crc16(buf, n) == _crc16_ref(data[:n])
    # classic check values: CRC-8 (poly 7) and CRC-16/BUYPASS (poly 0x8005, init 0) of "123456789"
    s = (C.c_uint8 * 9)(*b"123456789")
    assert lib.fo_crc8(s, 9) == 0xF4
    assert lib.fo_crc16(s, 9) == 0xFEE8


MD5_TARGETS = {  # (channels, bytes_per_sample) -> digest, src/test_libFLAC/md5.c:96-163
    (1, 1): "c19a5beb578f26ebfb347cef04316d7d", (1, 2): "d47890d3a9174e76ca4d272098368b2e",
    (1, 3): "5a4bd6aca17084197c0dfb5ba97bcb54", (1, 4): "79d57a32060bfe46a3e7bac5f7486f50",
    (2, 1): "89accf91f18ceaab461274bc4e82be7d", (2, 2): "b917165bd81cc84e5a28fbba87747644",
    (2, 3): "ec6392ca4f6b9eb19fec3b2c1530fd2a", (2, 4): "054dfdb89d8aa2dd2647c6fb4f23676d",
    (8, 2): "9d048fa4ea10ecb8a388e25d3ce2fb94", (8, 3): "5ad3d2756afaa742f3bf0ebc902af85f",
}


def test_md5_reference_vectors():
    h = engine.load_host()
    h.flacgpu_host_md5_init.argtypes = [C.c_void_p]
    h.flacgpu_host_md5_pcm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32]
    h.flacgpu_host_md5_final.argtypes = [C.c_void_p, C.c_void_p]
    h.flacgpu_host_md5_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    seed = 0x12345679
    arrays = np.zeros((8, 64), dtype=np.int64)
    for ch in range(8):                       # md5.c:175-183: LCG fills channel after channel
        for k in range(64):
            seed = (seed * 1103515245 + 12345) & 0xFFFFFFFF
            arrays[ch, k] = seed
    arrays = arrays.astype(np.uint32).view(np.int32)
    ctx = C.create_string_buffer(128)
    dig = C.create_string_buffer(16)
    for (ch, bs), want in MD5_TARGETS.items():
        inter = np.ascontiguousarray(arrays[:ch].T)
        h.flacgpu_host_md5_init(ctx)
        h.flacgpu_host_md5_pcm(ctx, inter.ctypes.data, ch, 64, bs)
        h.flacgpu_host_md5_final(ctx, dig)
        assert dig.raw.hex() == want, (ch, bs)
    # empty input digest (md5.c:58) and agreement with hashlib on odd chunkings
    h.flacgpu_host_md5_init(ctx)
    h.flacgpu_host_md5_final(ctx, dig)
    assert dig.raw.hex() == "d41d8cd98f00b204e9800998ecf8427e"
    rng = np.random.default_rng(5)
    blob = rng.integers(0, 256, 100000, dtype=np.uint8).tobytes()
    h.flacgpu_host_md5_init(ctx)
    pos = 0
    for step in (1, 63, 64, 65, 1000, 55, 56, 57, 119, 120, 121, 4096):
        h.flacgpu_host_md5_update(ctx, blob[pos:pos + step], step)
        pos += step
    h.flacgpu_host_md5_update(ctx, blob[pos:], len(blob) - pos)
    h.flacgpu_host_md5_final(ctx, dig)
    assert dig.raw.hex() == hashlib.md5(blob).hexdigest()


def _utf8_expected(v):
    """FLAC's extended UTF-8 (bitwriter.c:832-877)"""
    if v < 0x80:
        return bytes([v])
    n = 2 if v < 0x800 else 3 if v < 0x10000 else 4 if v < 0x200000 else 5 if v < 0x4000000 else 6
    lead = (0xFF << (8 - n)) & 0xFF
    out = [lead | (v >> (6 * (n - 1)))]
    for k in range(n - 2, -1, -1):
        out.append(0x80 | ((v >> (6 * k)) & 0x3F))
    return bytes(out)


def test_frame_header_utf8_and_crc8():
    """Frame numbers across every UTF-8 length class; header CRC-8 must verify (framing.c:356-388)."""
    pcm = np.zeros((1152, 1), dtype=np.int32)
    for fn in (0, 0x7F, 0x80, 0x7FF, 0x800, 0xFFFF, 0x10000, 0x1FFFFF, 0x200000, 0x3FFFFFF, 0x4000000, 0x7FFFFFFF):
        o = po.oracle_encode(pcm, 16, 44100, 0, first_frame=fn)
        d = o["data"]
        assert d[:2] == b"\xff\xf8"
        u = _utf8_expected(fn)
        assert d[4:4 + len(u)] == u
        hdr_len = 4 + len(u)
        assert _crc8_ref(d[:hdr_len]) == d[hdr_len]
        assert _crc16_ref(d[:-2]) == int.from_bytes(d[-2:], "big")


def test_the_residual_width_check_needs_no_logarithm_far_below_its_threshold():
    """ff_kernel (flacgpu_kernels.hip) skips fixed.c:299's logarithm when the error sum e is below n4 * 2^(sbps-1): there
    log2(ln2 * e / n4) < sbps - 1.5, so the float it becomes cannot reach sbps whatever the rounding.  Pinned here on the boundary
    of every width the kernel sees (n4 = 1148: a 1152-sample block)."""
    import math
    import numpy as np
    n4 = 1148
    for sbps in range(1, 18):
        for e in (1, (n4 << (sbps - 1)) // 2, (n4 << (sbps - 1)) - 1):
            if e < 1:
                continue
            rbps = np.float32(math.log(0.69314718055994530942 * e / n4) / 0.69314718055994530942)
            assert rbps < np.float32(sbps) - np.float32(1.4), (sbps, e, rbps)
        # and at twice the bound the estimate is still below the width: the kernel's exact branch decides there
        e2 = n4 << sbps
        assert np.float32(math.log(0.69314718055994530942 * e2 / n4) / 0.69314718055994530942) < np.float32(sbps)
