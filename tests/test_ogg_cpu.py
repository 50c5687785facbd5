"""CPU: the Ogg paging restatement of flac_amd/csrc/host/ogg.c (libogg's ogg_stream_packetin / pageout / flush, a library that
is not part of the reference tree) against Ogg FLAC streams the reference wrote WITH libogg: logical streams cut out of the
reference's fuzzing seed corpus (tests/golden/ogg/, make_vectors.py).  Their packets, queued and paged the way
ogg_encoder_aspect.c does it (metadata packets flushed, audio packets paged out, the last one closing the stream), must come
back as the same pages, byte for byte -- page boundaries, flags, granule positions, sequence numbers, CRC-32."""
import ctypes as C
import glob
import os

import pytest

from flac_amd import engine

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ogg", "oggflac_*.bin")))


class OggStream(C.Structure):
    _fields_ = [("body", C.c_void_p), ("body_storage", C.c_size_t), ("body_fill", C.c_size_t), ("body_returned", C.c_size_t),
                ("lacing", C.c_void_p), ("granule", C.c_void_p), ("lacing_storage", C.c_size_t), ("lacing_fill", C.c_size_t),
                ("header", C.c_uint8 * 282), ("header_len", C.c_size_t), ("e_o_s", C.c_int), ("b_o_s", C.c_int),
                ("serialno", C.c_long), ("pageno", C.c_long), ("packetno", C.c_int64), ("granulepos", C.c_int64)]


def _pages(d):
    pos = 0
    while pos < len(d):
        nseg = d[pos + 26]
        segs = list(d[pos + 27:pos + 27 + nseg])
        n = 27 + nseg + sum(segs)
        yield d[pos:pos + n], segs, d[pos + 27 + nseg:pos + n]
        pos += n


def _packets(d):
    out, cur = [], b""
    for _, segs, body in _pages(d):
        off = 0
        for s in segs:
            cur += body[off:off + s]
            off += s
            if s < 255:
                out.append(cur)
                cur = b""
    assert cur == b""
    return out


def _frame_blocksize(fr):
    """block size of a FLAC frame from its header (the 4-bit code, with the 8- / 16-bit field behind the UTF-8 frame number)"""
    code = fr[2] >> 4
    if code == 1:
        return 192
    if 2 <= code <= 5:
        return 576 << (code - 2)
    if code >= 8:
        return 256 << (code - 8)
    b = fr[4]
    n = 1 if b < 0x80 else 2 if b < 0xe0 else 3 if b < 0xf0 else 4 if b < 0xf8 else 5 if b < 0xfc else 6 if b < 0xfe else 7
    pos = 4 + n
    return fr[pos] + 1 if code == 6 else ((fr[pos] << 8) | fr[pos + 1]) + 1


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_paging_reproduces_libogg_streams(path):
    want = open(path, "rb").read()
    lib = engine.load_host()
    for n in ("fgh_ogg_stream_init", "fgh_ogg_stream_packetin", "fgh_ogg_stream_pageout", "fgh_ogg_stream_flush"):
        getattr(lib, n).restype = C.c_int
    lib.fgh_ogg_stream_packetin.argtypes = [C.POINTER(OggStream), C.c_char_p, C.c_size_t, C.c_int64, C.c_int]
    serial = int.from_bytes(want[14:18], "little", signed=True)
    os_ = OggStream()
    assert lib.fgh_ogg_stream_init(C.byref(os_), C.c_long(serial)) == 0
    packets = _packets(want)
    nmeta = 0                                   # header packets: the first one, then metadata blocks up to the is_last flag
    while True:
        pk = packets[nmeta]
        last = bool((pk[13] if nmeta == 0 else pk[0]) & 0x80)
        nmeta += 1
        if last:
            break
    got = b""
    samples = 0
    for i, pk in enumerate(packets):
        is_meta = i < nmeta
        if not is_meta:
            samples += _frame_blocksize(pk)
        assert lib.fgh_ogg_stream_packetin(C.byref(os_), pk, len(pk), samples, 1 if i + 1 == len(packets) else 0) == 0
        body, blen = C.c_void_p(), C.c_size_t()
        fn = lib.fgh_ogg_stream_flush if is_meta else lib.fgh_ogg_stream_pageout
        while fn(C.byref(os_), C.byref(body), C.byref(blen)):
            got += bytes(os_.header[:os_.header_len]) + C.string_at(body.value, blen.value)
    lib.fgh_ogg_stream_clear(C.byref(os_))
    assert len(FIXTURES) >= 3
    assert got == want
