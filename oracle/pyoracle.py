"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * oracle/liboracle.so        (our CPU restatement, oracle/flac_oracle.c)
  * oracle/_ref/libFLAC_ref.so (the unmodified reference libFLAC + oracle/ref_shim.c)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libFLAC_ref.so")

# preset table, stream_encoder.c:117-140:
# (mid_side, loose, max_lpc_order, max_partition_order, apodization)
PRESETS = {
    0: (0, 0, 0, 3, ("tukey", 0.5)),
    1: (1, 1, 0, 3, ("tukey", 0.5)),
    2: (1, 0, 0, 3, ("tukey", 0.5)),
    3: (0, 0, 6, 4, ("tukey", 0.5)),
    4: (1, 1, 8, 4, ("tukey", 0.5)),
    5: (1, 0, 8, 5, ("tukey", 0.5)),
    6: (1, 0, 8, 6, ("subdivide_tukey", 2)),
    7: (1, 0, 12, 6, ("subdivide_tukey", 2)),
    8: (1, 0, 12, 6, ("subdivide_tukey", 3)),
}


def build(ref=True):
    """Compile the oracle (and the reference, when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src/libFLAC"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "-j8"])


# ----------------------------------------------------------------------------- reference
class RefCfg(C.Structure):
    _fields_ = [
        ("channels", C.c_uint32), ("bps", C.c_uint32), ("sample_rate", C.c_uint32),
        ("level", C.c_int32), ("blocksize", C.c_uint32), ("do_md5", C.c_int32),
        ("num_threads", C.c_int32), ("disable_isa_mask", C.c_int32),
        ("limit_min_bitrate", C.c_int32), ("streamable_subset", C.c_int32),
        ("max_lpc_order", C.c_int32), ("qlp_precision", C.c_int32),
        ("min_partition_order", C.c_int32), ("max_partition_order", C.c_int32),
        ("mid_side", C.c_int32), ("loose_mid_side", C.c_int32),
        ("apodization", C.c_char_p),
        ("exhaustive", C.c_int32), ("prec_search", C.c_int32),
        ("disable_constant", C.c_int32), ("disable_fixed", C.c_int32), ("disable_verbatim", C.c_int32),
    ]


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_encode_stream.restype = C.c_int64
        lib.ref_encode_stream.argtypes = [C.POINTER(RefCfg), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                          C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_double)]
        lib.ref_encode_file.restype = C.c_int32
        lib.ref_encode_file.argtypes = [C.POINTER(RefCfg), C.c_void_p, C.c_uint64, C.c_char_p]
        lib.ref_vendor_string.restype = C.c_char_p
        _ref = lib
    return _ref


def ref_cfg(channels, bps, rate, level, blocksize=0, do_md5=0, num_threads=0, limit_min_bitrate=0,
            streamable_subset=1, max_lpc_order=-1, qlp_precision=-1, min_po=-1, max_po=-1,
            mid_side=-1, loose_mid_side=-1, apodization=None, disable_isa_mask=0, exhaustive=0, prec_search=0,
            disable=(0, 0, 0)):
    """disable = (constant, fixed, verbatim) subframes switched off"""
    return RefCfg(channels, bps, rate, level, blocksize, do_md5, num_threads, disable_isa_mask,
                  limit_min_bitrate, streamable_subset, max_lpc_order, qlp_precision, min_po, max_po,
                  mid_side, loose_mid_side, apodization.encode() if apodization else None, exhaustive, prec_search,
                  disable[0], disable[1], disable[2])


def ref_encode(pcm, bps, rate, level, want_bytes=True, **kw):
    """pcm: int32 array [nsamples, channels]. Returns dict(data, frame_bytes, header_bytes, seconds)."""
    lib = load_ref()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    cfg = ref_cfg(ch, bps, rate, level, **kw)
    cap = n * ch * 4 + 65536 + (n // 16) * 16
    out = np.empty(cap if want_bytes else 1, dtype=np.uint8)
    maxframes = n // 16 + 2
    fb = np.zeros(maxframes, dtype=np.uint32)
    nfr = C.c_uint32(0)
    hdr = C.c_uint64(0)
    el = C.c_double(0)
    r = lib.ref_encode_stream(C.byref(cfg), pcm.ctypes.data, n, out.ctypes.data if want_bytes else None, cap,
                              fb.ctypes.data, maxframes, C.byref(nfr), C.byref(hdr), C.byref(el))
    if r < 0:
        raise RuntimeError("reference encode failed: %d" % r)
    return dict(data=out[:r].tobytes() if want_bytes else None, total_bytes=int(r),
                frame_bytes=fb[:nfr.value].copy(), header_bytes=int(hdr.value), seconds=el.value)


def ref_encode_file(pcm, bps, rate, level, path, **kw):
    lib = load_ref()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    cfg = ref_cfg(ch, bps, rate, level, **kw)
    r = lib.ref_encode_file(C.byref(cfg), pcm.ctypes.data, n, path.encode())
    if r != 0:
        raise RuntimeError("reference file encode failed: %d" % r)
    with open(path, "rb") as f:
        return f.read()


# ------------------------------------------------------------------------------- oracle
FO_MAX_LPC_ORDER = 32


class FoApod(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("parts", C.c_uint32), ("window", C.POINTER(C.c_float))]


class FoConfig(C.Structure):
    _fields_ = [
        ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("sample_rate", C.c_uint32),
        ("blocksize", C.c_uint32), ("do_mid_side", C.c_uint32), ("loose_mid_side", C.c_uint32),
        ("max_lpc_order", C.c_uint32), ("qlp_coeff_precision", C.c_uint32),
        ("min_partition_order", C.c_uint32), ("max_partition_order", C.c_uint32),
        ("num_apodizations", C.c_uint32), ("apodizations", FoApod * 32),
        ("autoc_variant", C.c_uint32),
        ("disable_constant", C.c_uint32), ("disable_fixed", C.c_uint32), ("disable_verbatim", C.c_uint32),
        ("limit_min_bitrate", C.c_uint32),
        ("exhaustive", C.c_uint32), ("prec_search", C.c_uint32),
    ]


class FoSubframeInfo(C.Structure):
    _fields_ = [("type", C.c_uint32), ("order", C.c_uint32), ("wasted_bits", C.c_uint32), ("bits", C.c_uint32),
                ("partition_order", C.c_uint32), ("rice2", C.c_uint32), ("precision", C.c_uint32),
                ("shift", C.c_int32), ("qlp", C.c_int32 * 32)]


class FoFrameInfo(C.Structure):
    _fields_ = [("channel_assignment", C.c_uint32), ("frame_bytes", C.c_uint32), ("sub", FoSubframeInfo * 8)]


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        lib = C.CDLL(ORACLE_SO)
        lib.fo_encode_frame.restype = C.c_int64
        lib.fo_encode_frame.argtypes = [C.POINTER(FoConfig), C.POINTER(C.c_void_p), C.c_uint64, C.c_void_p,
                                        C.c_size_t, C.POINTER(FoFrameInfo)]
        lib.fo_encode_frames.restype = C.c_int64
        lib.fo_encode_frames.argtypes = [C.POINTER(FoConfig), C.POINTER(FoConfig), C.POINTER(C.c_void_p),
                                         C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.POINTER(C.c_uint32)]
        lib.fo_window_tukey.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        lib.fo_autocorrelation.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.fo_lp_coefficients.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        lib.fo_best_order.restype = C.c_uint32
        lib.fo_best_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.fo_quantize_coefficients.restype = C.c_int
        lib.fo_quantize_coefficients.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
        lib.fo_fixed_best_predictor.restype = C.c_uint32
        lib.fo_fixed_best_predictor.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.fo_fixed_best_predictor_ex.restype = C.c_uint32
        lib.fo_fixed_best_predictor_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        lib.fo_rice_search.restype = C.c_uint32
        lib.fo_rice_search.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
        lib.fo_crc8.restype = C.c_uint8
        lib.fo_crc8.argtypes = [C.c_void_p, C.c_size_t]
        lib.fo_crc16.restype = C.c_uint16
        lib.fo_crc16.argtypes = [C.c_void_p, C.c_size_t]
        lib.fo_expected_bits_per_residual_sample.restype = C.c_double
        lib.fo_expected_bits_per_residual_sample.argtypes = [C.c_double, C.c_uint32]
        _oracle = lib
    return _oracle


def default_qlp_precision(bps, blocksize):
    """stream_encoder.c:764-795"""
    if bps < 16:
        return max(5, 2 + bps // 2)
    if bps == 16:
        for lim, p in ((192, 7), (384, 8), (576, 9), (1152, 10), (2304, 11), (4608, 12)):
            if blocksize <= lim:
                return p
        return 13
    if blocksize <= 384:
        return 13
    if blocksize <= 1152:
        return 14
    return 15


def tukey_window(L, p):
    lib = load_oracle()
    w = np.empty(L, dtype=np.float32)
    lib.fo_window_tukey(w.ctypes.data, L, C.c_float(p))
    return w


class OracleConfig:
    """Keeps the numpy window tables alive next to the ctypes struct."""

    def __init__(self, channels, bps, rate, level, blocksize=None, stream_blocksize=None, limit_min_bitrate=0,
                 max_lpc_order=None, max_po=None, min_po=0, mid_side=None, loose=None, apod=None,
                 qlp_precision=None, exhaustive=0, prec_search=0, disable=(0, 0, 0)):
        ms, lo, lpc, mpo, ap = PRESETS[level]
        if max_lpc_order is not None:
            lpc = max_lpc_order
        if max_po is not None:
            mpo = max_po
        if mid_side is not None:
            ms = mid_side
        if loose is not None:
            lo = loose
        if apod is not None:
            ap = apod
        if stream_blocksize is None:
            stream_blocksize = blocksize if blocksize else (1152 if lpc == 0 else 4096)
        if blocksize is None:
            blocksize = stream_blocksize
        if channels != 2:
            ms = lo = 0
        if not ms:
            lo = 0
        self.stream_blocksize = stream_blocksize
        c = FoConfig()
        c.channels, c.bits_per_sample, c.sample_rate, c.blocksize = channels, bps, rate, blocksize
        c.do_mid_side, c.loose_mid_side, c.max_lpc_order = ms, lo, lpc
        # precision is resolved once per stream from the STREAM blocksize
        c.qlp_coeff_precision = qlp_precision or default_qlp_precision(bps, stream_blocksize)
        c.min_partition_order, c.max_partition_order = min_po, mpo
        c.num_apodizations = 1
        self.windows = []
        if ap[0] == "tukey":
            w = tukey_window(blocksize, ap[1])
            c.apodizations[0].kind, c.apodizations[0].parts = 0, 0
        else:
            parts = ap[1]
            w = tukey_window(blocksize, np.float32(np.float32(0.5) / np.float32(parts)))
            c.apodizations[0].kind, c.apodizations[0].parts = 1, parts
        self.windows.append(w)
        c.apodizations[0].window = w.ctypes.data_as(C.POINTER(C.c_float))
        c.autoc_variant = 8 if lpc < 8 else 12 if lpc < 12 else 16 if lpc < 16 else 0
        c.limit_min_bitrate = limit_min_bitrate
        c.exhaustive, c.prec_search = exhaustive, prec_search
        c.disable_constant, c.disable_fixed, c.disable_verbatim = disable
        self.c = c


def oracle_encode(pcm, bps, rate, level, first_frame=0, **kw):
    """pcm int32 [nsamples, channels] -> dict(data, frame_bytes) : audio frames only."""
    lib = load_oracle()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    planar = np.ascontiguousarray(pcm.T)
    cfg = OracleConfig(ch, bps, rate, level, **kw)
    N = cfg.c.blocksize
    tail = n % N
    tail_cfg = OracleConfig(ch, bps, rate, level, **dict(kw, blocksize=tail, stream_blocksize=N)) if tail else None
    ptrs = (C.c_void_p * ch)(*[planar[i].ctypes.data for i in range(ch)])
    cap = n * ch * 4 + 65536 + (n // 16 + 1) * 32
    out = np.empty(cap, dtype=np.uint8)
    nfmax = n // N + 2
    fb = np.zeros(nfmax, dtype=np.uint32)
    nf = C.c_uint32(0)
    r = lib.fo_encode_frames(C.byref(cfg.c), C.byref(tail_cfg.c) if tail_cfg else None, ptrs, n, first_frame,
                             out.ctypes.data, cap, fb.ctypes.data, C.byref(nf))
    if r < 0:
        raise RuntimeError("oracle encode failed: %d" % r)
    return dict(data=out[:r].tobytes(), frame_bytes=fb[:nf.value].copy())
