"""Helpers of the stream-decoder tests: the reference's decoder on a stream in memory (oracle/_ref/libFLAC_ref.so: ref_decode_stream,
oracle/ref_shim.c), the product's lane code + walk compiled for the host (oracle/libsdpin.so), streams written by the reference's own
`flac` tool (oracle/_ref/flac_cli_ref) and damage applied to them.  Test infrastructure."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libFLAC_ref.so")
PIN_SO = os.path.join(ROOT, "oracle", "libsdpin.so")
FLAC_REF = os.path.join(ROOT, "oracle", "_ref", "flac_cli_ref")

ERR_NAMES = {1: "LOST_SYNC", 2: "BAD_HEADER", 3: "FRAME_CRC_MISMATCH", 4: "UNPARSEABLE_STREAM", 5: "BAD_METADATA", 6: "OUT_OF_BOUNDS", 7: "MISSING_FRAME"}


def have_ref():
    return os.path.exists(REF_SO)


_ref = None
_pin = None


def _ref_lib():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.ref_decode_stream.restype = C.c_int
        _ref.ref_decode_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    return _ref


def _pin_lib():
    global _pin
    if _pin is None:
        if not os.path.exists(PIN_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        _pin = C.CDLL(PIN_SO)
        _pin.sdpin_decode.restype = C.c_int
        _pin.sdpin_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _pin.sdpin_probe.restype = C.c_int
        _pin.sdpin_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _pin


def ref_decode(stream, cap_samples=None, md5=True, max_events=4096):
    """The reference decoder on `stream` (bytes): dict(pcm [samples][channels] int32, events [status...], ok, md5_ok, state, ...)."""
    buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
    cap = cap_samples if cap_samples is not None else max(1 << 16, len(buf) * 16)
    pcm = np.zeros(cap * 8, dtype=np.int32)
    ev = np.zeros(max_events, dtype=np.uint32)
    res = np.zeros(16, dtype=np.uint64)
    rc = _ref_lib().ref_decode_stream(buf.ctypes.data, len(buf), 1 if md5 else 0, pcm.ctypes.data, cap, ev.ctypes.data, max_events, res.ctypes.data)
    assert rc == 0, rc
    assert not res[9], "reference output did not fit"
    ch = int(res[6]) or 1
    n = int(res[0])
    return dict(pcm=pcm[:n * ch].reshape(n, ch).copy(), events=[int(x) for x in ev[:min(int(res[2]), max_events)]], nevents=int(res[2]), ok=bool(res[3]),
                md5_ok=bool(res[4]), state=int(res[5]), channels=int(res[6]), bps=int(res[7]), frames=int(res[1]), format_changes=int(res[8]), samples=n)


def probe(stream):
    buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
    info = np.zeros(6, dtype=np.uint32)
    first = C.c_uint64(0)
    total = C.c_uint64(0)
    md5 = np.zeros(16, dtype=np.uint8)
    rc = _pin_lib().sdpin_probe(buf.ctypes.data, len(buf), info.ctypes.data, C.byref(first), C.byref(total), md5.ctypes.data)
    return rc, info, first.value, total.value, md5.tobytes()


def pin_decode(stream, cap_samples=None, max_events=4096, info=None, first_pos=None):
    """The product's lane code and walk on the host: the same dict as ref_decode (no MD5)."""
    buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
    rc, pinfo, pfirst, total, md5 = probe(stream)
    if info is None:
        info = pinfo
    if first_pos is None:
        first_pos = pfirst
    info = np.ascontiguousarray(info, dtype=np.uint32)
    cap = cap_samples if cap_samples is not None else max(1 << 16, len(buf) * 16)
    pcm = np.zeros(cap * 8, dtype=np.int32)
    ev = np.zeros(max_events, dtype=np.uint32)
    evp = np.zeros(max_events, dtype=np.uint64)
    res = np.zeros(16, dtype=np.uint64)
    rc = _pin_lib().sdpin_decode(buf.ctypes.data, len(buf), first_pos, info.ctypes.data, pcm.ctypes.data, cap, ev.ctypes.data, evp.ctypes.data, max_events, res.ctypes.data)
    assert rc == 0, rc
    ch = int(res[6]) or 1
    n = int(res[0])
    return dict(pcm=pcm[:n * ch].reshape(n, ch).copy(), events=[int(x) for x in ev[:min(int(res[3]), max_events)]], nevents=int(res[3]),
                event_pos=[int(x) for x in evp[:min(int(res[3]), max_events)]], ok=not bool(res[4]), channels=int(res[6]), bps=int(res[7]),
                frames=int(res[1]), silence=int(res[2]), format_changes=int(res[5]), samples=n, candidates=int(res[8]), retries=int(res[9]), long_rice_codes=int(res[10]),
                md5=md5, total_samples=total)


def flac_encode_cli(pcm, bps, rate, args, tool=FLAC_REF):
    """`flac <args>` of the reference's own tool on raw little-endian input -> the .flac file's bytes."""
    pcm = np.ascontiguousarray(pcm)
    n, ch = pcm.shape
    bytes_per = (bps + 7) // 8
    shifted = pcm.astype(np.int64)
    raw = np.zeros((n * ch, bytes_per), dtype=np.uint8)
    flat = shifted.reshape(-1)
    for k in range(bytes_per):
        raw[:, k] = (flat >> (8 * k)) & 0xff
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src = os.path.join(d, "in.raw")
        dst = os.path.join(d, "out.flac")
        raw.tofile(src)
        cmd = [tool, "--silent", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=%d" % ch, "--bps=%d" % bps,
               "--sample-rate=%d" % rate, "-f", "-o", dst] + list(args) + [src]
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.dirname(tool) + ":" + env.get("LD_LIBRARY_PATH", "")
        subprocess.check_call(cmd, env=env)
        return open(dst, "rb").read()


def same_verdict(a, b):
    """None when two decode results agree, else what differs."""
    if a["events"] != b["events"] or a["nevents"] != b["nevents"]:
        return "events %s vs %s" % ([ERR_NAMES.get(e, e) for e in a["events"][:12]], [ERR_NAMES.get(e, e) for e in b["events"][:12]])
    if a["ok"] != b["ok"]:
        return "ok %s vs %s" % (a["ok"], b["ok"])
    if a["samples"] != b["samples"] or a["channels"] != b["channels"]:
        return "samples %d x %d vs %d x %d" % (a["samples"], a["channels"], b["samples"], b["channels"])
    if a["format_changes"] != b["format_changes"]:
        return "format changes %d vs %d" % (a["format_changes"], b["format_changes"])
    if not np.array_equal(a["pcm"], b["pcm"]):
        d = np.argwhere(a["pcm"] != b["pcm"])[0]
        return "pcm differs first at sample %d channel %d: %d vs %d" % (d[0], d[1], a["pcm"][d[0], d[1]], b["pcm"][d[0], d[1]])
    return None


# ---- a FLAC frame writer for hand-built streams (format: SURVEY.md appendix B; src/libFLAC/stream_encoder_framing.c is what the
# reference writes, format.c:117-154 the field widths): frames the reference's encoder never emits -- escape-coded partitions, sample
# numbers and changing block sizes, any predictor, values that overflow -- with correct CRCs, so that only the decoder is on trial ----
class BitW:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def write(self, v, bits):
        if bits:
            self.acc = (self.acc << bits) | (int(v) & ((1 << bits) - 1))
            self.n += bits

    def unary(self, zeros):
        self.write(1, zeros + 1)

    def align(self):
        if self.n % 8:
            self.write(0, 8 - self.n % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.acc.to_bytes(self.n // 8, "big") if self.n else b""


def crc8(b):
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(b):
    c = 0
    for x in b:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out = []
    nb = 1
    while v >= (1 << (6 * nb + (6 - nb))):            # nb continuation bytes hold 6 nb + (6 - nb) bits
        nb += 1
    for k in range(nb):
        out.append(0x80 | ((v >> (6 * k)) & 0x3f))
    lead = (0xff << (7 - nb)) & 0xff | (v >> (6 * nb))
    return bytes([lead] + out[::-1])


FIXED_TAPS = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def subframe_bits(w, sf, n, nominal_bps):
    """sf: dict(kind, samples [n] (the coded channel, python ints), wasted=0, order, po=0, params=None | list per partition (int k, or
    ('esc', rawbits)), rice2=False, qlp=(coefs, precision, shift), raw_residual=None (residuals given instead of derived))"""
    kind = sf["kind"]
    wasted = sf.get("wasted", 0)
    x = [int(v) >> wasted for v in sf["samples"]]
    sb = nominal_bps - wasted
    code = {"constant": 0, "verbatim": 1}.get(kind)
    order = sf.get("order", 0)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 32 + order - 1
    w.write(0, 1); w.write(code, 6); w.write(1 if wasted else 0, 1)
    if wasted:
        w.unary(wasted - 1)
    if kind == "constant":
        w.write(x[0], sb); return
    if kind == "verbatim":
        for v in x:
            w.write(v, sb)
        return
    for v in x[:order]:
        w.write(v, sb)
    if kind == "lpc":
        coefs, prec, shift = sf["qlp"]
        w.write(prec - 1, 4); w.write(shift, 5)
        for c in coefs:
            w.write(c, prec)
        taps = list(coefs)
    else:
        taps, shift = FIXED_TAPS[order], 0
    res = sf.get("raw_residual")
    if res is None:
        res = []
        for i in range(order, n):
            pred = sum(taps[j] * x[i - 1 - j] for j in range(order)) >> shift
            res.append(x[i] - pred)
    rice2 = sf.get("rice2", False)
    po = sf.get("po", 0)
    w.write(1 if rice2 else 0, 2); w.write(po, 4)
    plen, esc = (5, 31) if rice2 else (4, 15)
    psize = n >> po
    params = sf.get("params") or [None] * (1 << po)
    pos = 0
    for p in range(1 << po):
        cnt = psize - order if p == 0 else psize
        if po == 0:
            cnt = n - order
        part = res[pos:pos + cnt]; pos += cnt
        k = params[p]
        if k is None:
            m = (sum(abs(r) for r in part) // max(len(part), 1)) if part else 0
            k = min(max(m.bit_length(), 0), esc - 1)
        if k == "esc":                                   # as many raw bits as this partition's residuals need
            need = max([(r.bit_length() if r >= 0 else (-r - 1).bit_length()) + 1 for r in part] + [1])
            k = ("esc", 0 if all(r == 0 for r in part) else need)
        if isinstance(k, tuple):
            w.write(esc, plen); w.write(k[1], 5)
            for r in part:
                w.write(r, k[1])
        else:
            w.write(k, plen)
            for r in part:
                u = (r << 1) if r >= 0 else ((-r) << 1) - 1
                w.unary(u >> k); w.write(u & ((1 << k) - 1), k)


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def build_frame(n, rate, bps, number, subframes, ca=0, variable=False, bs_code=None, sr_code=None, bps_code=None, reserved1=0, reserved2=0,
                break_crc8=False, break_crc16=False):
    """One frame.  subframes: a list of subframe dicts (see subframe_bits), in coded-channel order; ca: 0 independent, 1 left/side,
    2 right/side, 3 mid/side."""
    ch = len(subframes)
    w = BitW()
    w.write(0x3ffe, 14); w.write(reserved1, 1); w.write(1 if variable else 0, 1)
    tail = BitW()
    if bs_code is None:
        bs_code = BS_CODES.get(n)
        if bs_code is None:
            bs_code = 6 if n <= 256 else 7
    if bs_code == 6:
        tail.write(n - 1, 8)
    elif bs_code == 7:
        tail.write(n - 1, 16)
    if sr_code is None:
        sr_code = SR_CODES.get(rate)
        if sr_code is None:
            sr_code = 12 if rate % 1000 == 0 and rate // 1000 < 256 else 13 if rate < 65536 else 14
    if sr_code == 12:
        tail.write(rate // 1000, 8)
    elif sr_code == 13:
        tail.write(rate, 16)
    elif sr_code == 14:
        tail.write(rate // 10, 16)
    w.write(bs_code, 4); w.write(sr_code, 4)
    w.write(ch - 1 if ca == 0 else 7 + ca, 4)
    w.write(BPS_CODES[bps] if bps_code is None else bps_code, 3); w.write(reserved2, 1)
    for b in utf8_number(number):
        w.write(b, 8)
    w.write(tail.acc, tail.n)
    hdr = w.bytes()
    c8 = crc8(hdr) ^ (0x55 if break_crc8 else 0)
    w.write(c8, 8)
    for c, sf in enumerate(subframes):
        side = (ca == 1 and c == 1) or (ca == 2 and c == 0) or (ca == 3 and c == 1)
        subframe_bits(w, sf, n, bps + (1 if side else 0))
    w.align()
    body = w.bytes()
    c16 = crc16(body) ^ (0x5555 if break_crc16 else 0)
    return body + c16.to_bytes(2, "big")


def streaminfo_header(min_bs, max_bs, rate, ch, bps, total=0, md5=bytes(16)):
    w = BitW()
    w.write(min_bs, 16); w.write(max_bs, 16); w.write(0, 24); w.write(0, 24); w.write(rate, 20); w.write(ch - 1, 3); w.write(bps - 1, 5); w.write(total, 36)
    return b"fLaC" + bytes([0x80, 0, 0, 34]) + w.bytes() + md5
