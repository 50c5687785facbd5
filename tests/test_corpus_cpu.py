"""CPU: the host side of the corpus job (flac_amd/corpus.py) -- the synthetic corpus as the host and the device generate it, the
track split, the stream header against the reference's own file, the threaded many-chain MD5 on the corpus' tracks, and the
command line's way to flac_amd.dist.ensure_ranks."""
import hashlib
import struct

import numpy as np
import pytest

from flac_amd import corpus as co


def test_host_and_device_generators_agree():
    """host_frames (numpy) and device_frames (torch, here on the CPU) produce the same samples for any frame range, and host_corpus
    is host_frames of the whole corpus"""
    import torch
    base = co.base_clip()
    base_t = torch.from_numpy(base)
    for f0, f1 in ((0, 3), (510, 515), (1023, 1030), (40 * 512 - 2, 40 * 512 + 3)):
        want = co.host_frames(base, f0, f1)
        out = torch.empty(((f1 - f0) * co.BLOCK, co.CH), dtype=torch.int16)
        co.device_frames(base_t, f0, f1, out)
        assert np.array_equal(out.numpy(), want), (f0, f1)


def test_track_ranges_cover_the_corpus_once():
    for F, n in ((387598, 120), (646, 11), (5, 8), (1000, 1000), (7, 1)):
        r = co.track_ranges(F, n)
        assert len(r) == n and r[0][0] == 0 and r[-1][1] == F
        assert all(r[i][1] == r[i + 1][0] for i in range(n - 1))
        sizes = [hi - lo for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_stream_header_is_what_the_reference_writes(ref, tmp_path):
    """"fLaC" + STREAMINFO + VORBIS_COMMENT of a corpus stream == the first bytes of the reference's own file for the same samples"""
    from oracle import pyoracle as po
    base = co.base_clip()
    pcm = co.host_frames(base, 0, 3)[: 2 * co.BLOCK + 777].astype(np.int32)
    want = po.ref_encode_file(pcm, 16, 44100, 8, str(tmp_path / "r.flac"), do_md5=1)
    o = po.oracle_encode(pcm, 16, 44100, 8)
    fb = np.asarray(o["frame_bytes"])
    md5 = hashlib.md5(pcm.astype("<i2").tobytes()).digest()
    hdr = co.stream_header(pcm.shape[0], int(fb.min()), int(fb.max()), md5)
    assert want[:len(hdr)] == hdr
    assert want[len(hdr):] == o["data"]


@pytest.mark.parametrize("ntracks,threads", [(7, 3), (40, 4), (120, 8)])
def test_tracks_hashed_from_one_buffer_by_threads(ntracks, threads):
    """md5_many_mt on pointers INTO one buffer (the job hashes its tracks straight from the pinned input): every digest = hashlib's"""
    base = co.base_clip()
    F = 200
    total = F * co.BLOCK - 1234
    buf = co.host_frames(base, 0, F)
    ranges = co.track_ranges(F, ntracks)
    ptrs = [buf[lo * co.BLOCK:].ctypes.data if hi > lo else 0 for lo, hi in ranges]
    lens = [max(0, min(hi * co.BLOCK, total) - lo * co.BLOCK) * co.CH * 2 if hi > lo else 0 for lo, hi in ranges]
    got = co.md5_many_mt(ptrs, lens, threads)
    for t, (lo, hi) in enumerate(ranges):
        want = hashlib.md5(buf[lo * co.BLOCK:min(hi * co.BLOCK, total)].tobytes()).digest() if hi > lo else hashlib.md5(b"").digest()
        assert got[t] == want, t


def test_usable_cpus_is_sane():
    n = co.usable_cpus()
    assert 1 <= n <= 4096
