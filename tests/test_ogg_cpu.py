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


# ---- packets that span many pages (the engine's largest frames: 8 channels x 65535 samples x 32 bits ~ 2.1 MB) -------------
# No libogg-written vector of that size exists in the reference tree, so the pager is held against a second, independent
# restatement of libogg 1.3's published framing algorithm (framing.c: ogg_stream_packetin / ogg_stream_flush_i), written here
# in Python, on random packet sequences; and every page is checked on its own (CRC-32, flags, sequence, lacing).
def _ogg_crc(data):
    if not hasattr(_ogg_crc, "t"):
        t = []
        for i in range(256):
            r = i << 24
            for _ in range(8):
                r = ((r << 1) ^ 0x04c11db7) & 0xffffffff if r & 0x80000000 else (r << 1) & 0xffffffff
            t.append(r)
        _ogg_crc.t = t
    c = 0
    for b in data:
        c = ((c << 8) & 0xffffffff) ^ _ogg_crc.t[((c >> 24) & 0xff) ^ b]
    return c


class PyPager:
    def __init__(self, serial):
        self.lacing, self.granule, self.body = [], [], bytearray()
        self.serial, self.pageno, self.bos, self.eos, self.granulepos = serial, 0, 0, 0, 0

    def packetin(self, pk, granulepos, eos):
        n = len(pk) // 255 + 1
        first = len(self.lacing)
        for _ in range(n - 1):
            self.lacing.append(255); self.granule.append(self.granulepos)
        self.lacing.append(len(pk) % 255); self.granule.append(granulepos)
        self.granulepos = granulepos
        self.lacing[first] |= 0x100
        self.body += pk
        if eos:
            self.eos = 1

    def _flush(self, force, nfill=4096):
        maxvals = min(len(self.lacing), 255)
        if maxvals == 0:
            return None
        acc, granule_pos = 0, -1
        if self.bos == 0:
            granule_pos = 0
            vals = 0
            while vals < maxvals:
                if (self.lacing[vals] & 0xff) < 255:
                    vals += 1
                    break
                vals += 1
        else:
            packets_done = just = 0
            vals = 0
            while vals < maxvals:
                if acc > nfill and just >= 4:
                    force = 1
                    break
                acc += self.lacing[vals] & 0xff
                if (self.lacing[vals] & 0xff) < 255:
                    granule_pos = self.granule[vals]
                    packets_done += 1
                    just = packets_done
                else:
                    just = 0
                vals += 1
            if vals == 255:
                force = 1
        if not force:
            return None
        flags = (0 if self.lacing[0] & 0x100 else 1) | (2 if self.bos == 0 else 0) | (4 if self.eos and len(self.lacing) == vals else 0)
        self.bos = 1
        hdr = bytearray(b"OggS\0") + bytes([flags]) + (granule_pos & 0xffffffffffffffff).to_bytes(8, "little")
        hdr += (self.serial & 0xffffffff).to_bytes(4, "little") + self.pageno.to_bytes(4, "little") + bytes(4) + bytes([vals])
        hdr += bytes(v & 0xff for v in self.lacing[:vals])
        self.pageno += 1
        nbody = sum(v & 0xff for v in self.lacing[:vals])
        body = bytes(self.body[:nbody])
        del self.body[:nbody]
        del self.lacing[:vals]
        del self.granule[:vals]
        crc = _ogg_crc(bytes(hdr) + body)
        hdr[22:26] = crc.to_bytes(4, "little")
        return bytes(hdr) + body

    def pageout(self):
        force = 1 if ((self.eos and self.lacing) or (self.lacing and not self.bos)) else 0
        return self._flush(force)

    def flush(self):
        return self._flush(1)


def _c_pager():
    lib = engine.load_host()
    for n in ("fgh_ogg_stream_init", "fgh_ogg_stream_packetin", "fgh_ogg_stream_pageout", "fgh_ogg_stream_flush"):
        getattr(lib, n).restype = C.c_int
    lib.fgh_ogg_stream_packetin.argtypes = [C.POINTER(OggStream), C.c_char_p, C.c_size_t, C.c_int64, C.c_int]
    return lib


@pytest.mark.parametrize("seed", range(6))
def test_packets_spanning_many_pages(seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    lib = _c_pager()
    serial = int(rng.integers(-2**31, 2**31))
    # header packets (flushed one by one, as ogg_encoder_aspect.c does), then audio packets paged out
    sizes_meta = [51, 40] + [int(rng.integers(1, 70000)) for _ in range(2)]
    big = [2 * 1024 * 1024 + 77, 255 * 255, 255 * 255 + 1, 255 * 255 - 1, 255 * 510, 0, 255, 254, 256]
    sizes = [int(rng.integers(1, 20000)) for _ in range(int(rng.integers(5, 40)))] + big + [int(rng.integers(14, 3000)) for _ in range(50)]
    order = rng.permutation(len(sizes))
    sizes = [sizes[i] for i in order]
    packets = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in sizes_meta + sizes]
    os_ = OggStream()
    assert lib.fgh_ogg_stream_init(C.byref(os_), C.c_long(serial)) == 0
    py = PyPager(serial)
    got, want = [], []
    gran = 0
    for i, pk in enumerate(packets):
        meta = i < len(sizes_meta)
        if not meta:
            gran += 4096
        eos = 1 if i + 1 == len(packets) else 0
        assert lib.fgh_ogg_stream_packetin(C.byref(os_), pk, len(pk), gran if not meta else 0, eos) == 0
        py.packetin(pk, gran if not meta else 0, eos)
        body, blen = C.c_void_p(), C.c_size_t()
        fn = lib.fgh_ogg_stream_flush if meta else lib.fgh_ogg_stream_pageout
        while fn(C.byref(os_), C.byref(body), C.byref(blen)):
            got.append(bytes(os_.header[:os_.header_len]) + C.string_at(body.value, blen.value))
        while True:
            pg = py.flush() if meta else py.pageout()
            if pg is None:
                break
            want.append(pg)
    lib.fgh_ogg_stream_clear(C.byref(os_))
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a == b
    stream = b"".join(got)
    # every page on its own, and the packets come back
    seq = 0
    for page, segs, body in _pages(stream):
        assert page[:5] == b"OggS\0" and len(segs) <= 255
        assert int.from_bytes(page[18:22], "little") == seq
        seq += 1
        z = bytearray(page); z[22:26] = bytes(4)
        assert _ogg_crc(bytes(z)) == int.from_bytes(page[22:26], "little")
    assert stream[5] & 2 and not stream[5] & 1                  # first page: beginning of stream, not a continuation
    assert _packets(stream) == packets
    # the 2 MB packet: its middle pages are 255 segments of 255 bytes with no granule position
    full = [p for p, segs, _ in _pages(stream) if len(segs) == 255 and all(s == 255 for s in segs)]
    assert len(full) >= 30
    assert all(int.from_bytes(p[6:14], "little", signed=True) == -1 for p in full)          # no packet ends on such a page
