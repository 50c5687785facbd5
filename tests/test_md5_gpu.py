"""-m gpu: the many-stream MD5 on the device (flacgpu_md5.hip, one lane per stream) against hashlib: RFC 1321's test suite, every
length around the padding boundaries (55 / 56 / 63 / 64 / 65 bytes ...), every byte alignment of a stream's start, ragged long
streams (lanes that finish at different times), and more streams than one wavefront."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(buffers, align_offsets=None):
    import torch
    from flac_amd import engine
    offs, pos = [], 0
    for i, b in enumerate(buffers):
        pos += 0 if align_offsets is None else align_offsets[i % len(align_offsets)]
        offs.append(pos)
        pos += len(b) + 3
    host = np.zeros(pos + 64, dtype=np.uint8)
    host[:] = 0xA5                                             # what lies around a stream must not leak into its digest
    for o, b in zip(offs, buffers):
        host[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    return engine.md5_many_device(d.data_ptr(), offs, [len(b) for b in buffers])


def test_rfc1321_suite():
    msgs = [b"", b"a", b"abc", b"message digest", b"abcdefghijklmnopqrstuvwxyz",
            b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", b"1234567890" * 8]
    got = _run(msgs)
    assert [g.hex() for g in got] == [hashlib.md5(m).hexdigest() for m in msgs]
    assert got[0].hex() == "d41d8cd98f00b204e9800998ecf8427e" and got[2].hex() == "900150983cd24fb0d6963f7d28e17f72"


def test_every_length_and_alignment_around_the_padding():
    rng = np.random.default_rng(7)
    bufs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in list(range(0, 200)) + [255, 256, 257, 1000, 4095, 4096, 4097]]
    for al in ([0], [1, 2, 3, 0, 5, 7]):
        got = _run(bufs, al)
        assert got == [hashlib.md5(b).digest() for b in bufs]


def test_ragged_long_streams_and_many_of_them():
    rng = np.random.default_rng(8)
    lens = [int(v) for v in rng.integers(0, 300000, 200)] + [1 << 20, (1 << 20) + 1]
    bufs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    assert _run(bufs, [0, 2]) == [hashlib.md5(b).digest() for b in bufs]
