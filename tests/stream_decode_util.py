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
