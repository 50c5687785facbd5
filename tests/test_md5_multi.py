"""CPU: the eight-chains-at-once MD5 of flac_amd/csrc/host/md5.c (AVX2, one chain per 32-bit lane) against hashlib and against the
reference library's FLAC__MD5* where oracle/_ref is built: RFC 1321's vectors and the reference's own (src/test_libFLAC/md5.c: the
zero-length message and 'a' repeated), buffers of unequal length, lengths around the block and padding boundaries."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import flac_amd


@pytest.fixture(scope="module")
def host():
    lib = flac_amd.engine.load_host()
    lib.flacgpu_host_md5_many.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_void_p]
    lib.flacgpu_host_md5_many.restype = None
    lib.flacgpu_host_md5_x8_available.restype = C.c_int
    return lib


def many(lib, bufs):
    n = len(bufs)
    keep = [np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8) for b in bufs]
    ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in keep])
    lens = (C.c_size_t * n)(*[len(b) for b in bufs])
    out = np.zeros((n, 16), dtype=np.uint8)
    lib.flacgpu_host_md5_many(ptrs, lens, n, out.ctypes.data)
    return [out[i].tobytes().hex() for i in range(n)]


def test_rfc1321_and_reference_vectors(host):
    msgs = [b"", b"a", b"abc", b"message digest", b"abcdefghijklmnopqrstuvwxyz", b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789",
            b"1234567890" * 8, b"a" * 1000000]
    want = ["d41d8cd98f00b204e9800998ecf8427e", "0cc175b9c0f1b6a831c399e269772661", "900150983cd24fb0d6963f7d28e17f72", "f96b697d7cb7938d525a2f31aaf161d0",
            "c3fcd3d76192e4007dfb496cca67e13b", "d174ab98d277d9f5a5611c2c9f419d9f", "57edf4a22be3c955ac49da2e2107b67a", "7707d6ae4e027c70eea2a935c2296f21"]
    assert many(host, msgs) == want                  # eight buffers: one pass of the eight-lane routine (where the CPU has AVX2)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 16, 23])
def test_unequal_lengths(host, n):
    rng = np.random.default_rng(n)
    lens = [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 4096, 4097, 100000, 3 * 64, 8191, 5, 64 * 1000, 999, 12345, 64, 1, 77777][:n]
    bufs = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
    assert many(host, bufs) == [hashlib.md5(b).hexdigest() for b in bufs]


def test_agrees_with_the_reference_library(host, ref, tmp_path):
    """FLAC__MD5Accumulate on 16-bit stereo blocks == our digest of the little-endian sample bytes (what STREAMINFO holds)"""
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    tracks = [rng.integers(-32768, 32768, size=(4096 * (3 + i) + 17 * i, 2), dtype=np.int64).astype(np.int32) for i in range(9)]
    ours = many(host, [t.astype("<i2").tobytes() for t in tracks])
    for t, d in zip(tracks, ours):
        data = po.ref_encode_file(t, 16, 44100, 0, str(tmp_path / "r.flac"), do_md5=1)
        assert data[8 + 18:8 + 34].hex() == d                   # STREAMINFO's MD5 field of the reference's own file


def test_rate_note(host, capsys):
    import time
    if not host.flacgpu_host_md5_x8_available():
        pytest.skip("no AVX2 on this host")
    rng = np.random.default_rng(1)
    bufs = [rng.integers(0, 256, 8 << 20, dtype=np.uint8).tobytes() for _ in range(8)]
    t0 = time.perf_counter(); many(host, bufs); t8 = time.perf_counter() - t0
    t0 = time.perf_counter(); many(host, bufs[:1]); t1 = time.perf_counter() - t0
    with capsys.disabled():
        print("\n[md5] one chain %.2f GB/s, eight chains at once %.2f GB/s" % ((8 << 20) / t1 / 1e9, 8 * (8 << 20) / t8 / 1e9))
    assert t8 < 8 * t1


def many_mt(lib, bufs, nthreads):
    lib.flacgpu_host_md5_many_mt.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_void_p, C.c_uint32]
    lib.flacgpu_host_md5_many_mt.restype = None
    n = len(bufs)
    keep = [np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8) for b in bufs]
    ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in keep])
    lens = (C.c_size_t * n)(*[len(b) for b in bufs])
    out = np.zeros((n, 16), dtype=np.uint8)
    lib.flacgpu_host_md5_many_mt(ptrs, lens, n, out.ctypes.data, nthreads)
    return [out[i].tobytes().hex() for i in range(n)]


@pytest.mark.parametrize("n,threads", [(1, 4), (15, 2), (16, 1), (17, 3), (33, 2), (40, 4), (120, 16), (120, 3), (64, 2)])
def test_sixteen_chains_and_host_threads(host, n, threads):
    """flacgpu_host_md5_many_mt: groups of sixteen chains (AVX-512: one vpternlogd per round function, vprold) where that still gives
    every thread a group, else eight (AVX2), else single chains; work items handed to `threads` threads -- every digest equal to
    hashlib's, for unequal lengths around the block and padding boundaries (a corpus of 120 tracks is the job of VERDICT r04 #6)"""
    rng = np.random.default_rng(100 + n)
    edge = [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 4096, 4097, 3 * 64, 8191]
    lens = [edge[i % len(edge)] + (int(rng.integers(0, 5000)) * 64 if i % 3 == 0 else int(rng.integers(0, 70000))) for i in range(n)]
    bufs = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for L in lens]
    want = [hashlib.md5(b).hexdigest() for b in bufs]
    assert many_mt(host, bufs, threads) == want
    assert many(host, bufs) == want                  # (the single-thread entry takes sixteen-wide groups, too)


def test_rate_note_sixteen(host, capsys):
    import time
    host.flacgpu_host_md5_x16_available.restype = C.c_int
    if not host.flacgpu_host_md5_x16_available():
        pytest.skip("no AVX-512 on this host")
    rng = np.random.default_rng(1)
    one = rng.integers(0, 256, 4 << 20, dtype=np.uint8).tobytes()
    bufs = [one] * 16
    t0 = time.perf_counter(); many(host, bufs[:8]); t8 = time.perf_counter() - t0
    t0 = time.perf_counter(); many(host, bufs); t16 = time.perf_counter() - t0
    with capsys.disabled():
        print("\n[md5] eight chains (AVX2) %.2f GB/s, sixteen chains (AVX-512) %.2f GB/s on one thread" % (8 * (4 << 20) / t8 / 1e9, 16 * (4 << 20) / t16 / 1e9))
