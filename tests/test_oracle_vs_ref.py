"""Pins the oracle (and the host C layer's window tables) to the REAL reference binary, stage by stage
and end to end.  Needs oracle/_ref/libFLAC_ref.so (built here from /root/reference; it travels to the
GPU box as a prebuilt file).  Skipped where it is absent -- tests/test_oracle_golden.py then carries
the pin through committed digests."""
import ctypes as C

import numpy as np
import pytest

import signals
from oracle import pyoracle as po
from flac_amd import engine


def _frames(r):
    return r["data"][r["header_bytes"]:]


@pytest.mark.parametrize("variant,name", [(8, "FLAC__lpc_compute_autocorrelation_intrin_fma_lag_8"),
                                          (12, "FLAC__lpc_compute_autocorrelation_intrin_fma_lag_12"),
                                          (16, "FLAC__lpc_compute_autocorrelation_intrin_fma_lag_16")])
def test_autocorrelation_association_order(ref, variant, name):
    """lpc_intrin_fma.c:46-72 as compiled: every lag bit for bit, many lengths."""
    fn = getattr(ref, name)
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    orc = po.load_oracle()
    rng = np.random.default_rng(42)
    for n in (4096, 2048, 1365, 1152, 576, 4100, 3392, 683, 100, 61, 52, 44, 36, 33):
        for trial in range(20):
            scale = 10.0 ** rng.integers(0, 5)
            d = (rng.standard_normal(n) * scale).astype(np.float32)
            if trial % 4 == 0:      # pure tone: the ill-conditioned case
                d = (np.sin(np.arange(n) * rng.uniform(0.01, 3.0)) * scale).astype(np.float32)
            a = np.zeros(33)
            b = np.zeros(33)
            fn(d.ctypes.data, n, variant, a.ctypes.data)
            orc.fo_autocorrelation(variant, d.ctypes.data, n, variant, b.ctypes.data)
            assert np.array_equal(a[:variant].view(np.uint64), b[:variant].view(np.uint64)), (variant, n, trial)


def test_levinson_order_quantise(ref):
    """lpc.c:176 (with the compiled (r+1)*lpc[j] factoring), :1608, :220"""
    orc = po.load_oracle()
    ref.FLAC__lpc_compute_lp_coefficients.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
    ref.FLAC__lpc_compute_best_order.restype = C.c_uint32
    ref.FLAC__lpc_compute_best_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ref.FLAC__lpc_quantize_coefficients.restype = C.c_int
    ref.FLAC__lpc_quantize_coefficients.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_int)]
    rng = np.random.default_rng(7)
    for trial in range(3000):
        n = 4096
        # AR-like signal -> autocorrelation
        x = rng.standard_normal(n + 40)
        k = rng.standard_normal(rng.integers(1, 6)) * 0.5
        x = np.convolve(x, np.concatenate([[1.0], k]))[:n] * 10 ** rng.uniform(0, 4)
        if trial % 5 == 0:
            x = np.sin(np.arange(n) * rng.uniform(0.01, 3)) * 1000
        order = int(rng.integers(1, 16))
        autoc = np.array([np.dot(x[j:], x[:n - j]) for j in range(order + 1)] + [0.0] * (33 - order - 1))
        mo_a, mo_b = C.c_uint32(order), C.c_uint32(order)
        lp_a = np.zeros((32, 32), np.float32); lp_b = np.zeros((32, 32), np.float32)
        er_a = np.zeros(32); er_b = np.zeros(32)
        ref.FLAC__lpc_compute_lp_coefficients(autoc.ctypes.data, C.byref(mo_a), lp_a.ctypes.data, er_a.ctypes.data)
        orc.fo_lp_coefficients(autoc.ctypes.data, C.byref(mo_b), lp_b.ctypes.data, er_b.ctypes.data)
        assert mo_a.value == mo_b.value
        m = mo_a.value
        assert np.array_equal(er_a[:m].view(np.uint64), er_b[:m].view(np.uint64)), trial
        assert np.array_equal(lp_a[:m].view(np.uint32), lp_b[:m].view(np.uint32)), trial
        assert ref.FLAC__lpc_compute_best_order(er_a.ctypes.data, m, n, 28) == orc.fo_best_order(er_b.ctypes.data, m, n, 28)
        for prec in (5, 9, 12, 15):
            qa = np.zeros(32, np.int32); qb = np.zeros(32, np.int32)
            sa, sb = C.c_int(0), C.c_int(0)
            ra = ref.FLAC__lpc_quantize_coefficients(lp_a[m - 1].ctypes.data, m, prec, qa.ctypes.data, C.byref(sa))
            rb = orc.fo_quantize_coefficients(lp_b[m - 1].ctypes.data, m, prec, qb.ctypes.data, C.byref(sb))
            assert ra == rb
            if ra == 0:
                assert sa.value == sb.value and np.array_equal(qa, qb)


def test_fixed_best_predictor_variants(ref):
    """fixed_intrin_ssse3.c:62 (narrow, exact) and fixed_intrin_avx2.c:57 (wide, lane quirk for n%4 != 0)"""
    orc = po.load_oracle()
    for name, wide in (("FLAC__fixed_compute_best_predictor_intrin_ssse3", 0), ("FLAC__fixed_compute_best_predictor_wide_intrin_avx2", 1)):
        fn = getattr(ref, name)
        fn.restype = C.c_uint32
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        rng = np.random.default_rng(11 + wide)
        for n in list(range(1, 40)) + [119, 1148, 1149, 1150, 1151, 4092, 4091, 2045]:
            for trial in range(8):
                amp = 1 << int(rng.integers(1, 15 if not wide else 23))
                x = rng.integers(-amp, amp, n + 4).astype(np.int32)
                if trial == 0:
                    x[:] = 5
                ra = np.zeros(5, np.float32); rb = np.zeros(5, np.float32)
                oa = fn(x.ctypes.data + 16, n, ra.ctypes.data)
                ob = orc.fo_fixed_best_predictor_ex(x.ctypes.data + 16, n, rb.ctypes.data, wide)
                assert oa == ob, (name, n, trial)
                assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)), (name, n, trial, ra, rb)


WINDOW_SPECS = ["bartlett", "bartlett_hann", "blackman", "blackman_harris_4term_92db", "connes", "flattop", "gauss(0.2)",
                "hamming", "hann", "kaiser_bessel", "nuttall", "rectangle", "triangle", "tukey(0.5)", "tukey(0.25)",
                "partial_tukey(2)", "partial_tukey(3/0.3/0.5)", "punchout_tukey(3)", "punchout_tukey(2/0.2/0.4)",
                "subdivide_tukey(3)", "subdivide_tukey(2/0.7)", "welch"]


@pytest.mark.parametrize("spec", WINDOW_SPECS)
def test_host_window_tables_and_apodization_parser(ref, spec):
    """flac_amd/csrc/host/{window,settings}.c vs window.c + set_apodization: encode with the spec through the
    reference and through the oracle fed with the HOST layer's window tables -- frames must match."""
    pcm = signals.music(4096 * 2 + 333, 2, 16, seed=3)
    r = po.ref_encode(pcm, 16, 44100, 5, apodization=spec, max_lpc_order=8)
    s = engine.make_settings(2, 16, 44100, 5, apodization=spec, max_lpc_order=8)
    frames = b""
    for blk, off, fn in ((4096, 0, 0), (4096, 4096, 1), (333, 8192, 2)):
        cfg = po.OracleConfig(2, 16, 44100, 5, blocksize=blk, stream_blocksize=4096, max_lpc_order=8)
        w = engine.host_windows(s, blk)
        cfg.c.num_apodizations = s.num_apodizations
        for a in range(s.num_apodizations):
            sub = s.apodizations[a].type == 16
            cfg.c.apodizations[a].kind = 1 if sub else 0
            cfg.c.apodizations[a].parts = s.apodizations[a].parts if sub else 0
            cfg.c.apodizations[a].window = w[a].ctypes.data_as(C.POINTER(C.c_float))
        planar = np.ascontiguousarray(pcm[off:off + blk].T)
        ptrs = (C.c_void_p * 2)(planar[0].ctypes.data, planar[1].ctypes.data)
        out = np.empty(65536, np.uint8)
        n = po.load_oracle().fo_encode_frame(C.byref(cfg.c), ptrs, fn, out.ctypes.data, out.size, None)
        assert n > 0
        frames += out[:n].tobytes()
    assert frames == _frames(r)


@pytest.mark.parametrize("level", range(9))
@pytest.mark.parametrize("family", ["music", "white", "sine", "mixed", "wasted", "square"])
def test_streams_16bit(ref, family, level):
    pcm = signals.FAMILIES[family](4096 * 5 + 777, 2, 16)
    r = po.ref_encode(pcm, 16, 44100, level)
    o = po.oracle_encode(pcm, 16, 44100, level)
    assert o["data"] == _frames(r)
    assert np.array_equal(o["frame_bytes"], r["frame_bytes"])


@pytest.mark.parametrize("level", [0, 2, 5, 8])
def test_streams_24bit_96k(ref, level):
    for fam in ("music", "white", "sine", "wasted"):
        pcm = signals.FAMILIES[fam](4096 * 3 + 123, 2, 24)
        r = po.ref_encode(pcm, 24, 96000, level)
        assert po.oracle_encode(pcm, 24, 96000, level)["data"] == _frames(r)


@pytest.mark.parametrize("tail", [1, 2, 3, 4, 5, 6, 16, 31, 32, 33, 34, 63, 65, 100, 255, 257, 1000, 1931, 1932, 2048, 3859, 3860, 4095])
def test_short_last_block(ref, tail):
    for level in (0, 5, 8):
        n = (1152 if level < 3 else 4096) + tail
        for bps, rate in ((16, 44100), (24, 96000)):
            pcm = signals.music(n, 2, bps, seed=tail)
            r = po.ref_encode(pcm, bps, rate, level)
            assert po.oracle_encode(pcm, bps, rate, level)["data"] == _frames(r), (tail, level, bps)


def test_pure_tones(ref):
    """10/49 frames of the reference's own sine16-16 stream change if the summation order is wrong."""
    for f in (55.5, 441.0, 997.0, 1000.0, 4410.0, 11025.0, 20000.0):
        for level in (3, 5, 8):
            pcm = signals.sine(4096 * 8 + 777, 1, 16, freq=f)
            r = po.ref_encode(pcm, 16, 44100, level)
            assert po.oracle_encode(pcm, 16, 44100, level)["data"] == _frames(r), (f, level)


def test_limit_min_bitrate_and_channels(ref):
    for level in (0, 2, 5, 8):
        for pcm in (signals.silence(4096 * 3, 2, 16), signals.mixed(4096 * 6, 2, 16), signals.silence(4096 * 2, 1, 16)):
            r = po.ref_encode(pcm, 16, 44100, level, limit_min_bitrate=1)
            assert po.oracle_encode(pcm, 16, 44100, level, limit_min_bitrate=1)["data"] == _frames(r)
    for ch in (1, 3, 4, 6, 8):
        pcm = signals.music(4096 * 2 + 50, ch, 16, seed=ch)
        r = po.ref_encode(pcm, 16, 48000, 8)
        assert po.oracle_encode(pcm, 16, 48000, 8)["data"] == _frames(r)


@pytest.mark.parametrize("mode", [(1, 0), (0, 1), (1, 1)], ids=["e", "p", "ep"])
@pytest.mark.parametrize("level", [3, 5, 8])
def test_exhaustive_and_precision_search(ref, level, mode):
    """-e (every fixed order, every LPC order) and -p (every coefficient precision), stream_encoder.c:4155-4243."""
    ex, ps = mode
    cases = [("music", 16, 44100), ("mixed", 16, 44100), ("sine", 16, 44100)]
    if level == 8:
        cases.append(("music", 24, 96000))
    for fam, bps, rate in cases:
        n = 4096 * 2 + 411 if not (ex and ps) else 4096 + 411
        pcm = signals.FAMILIES[fam](n, 2, bps)
        r = po.ref_encode(pcm, bps, rate, level, exhaustive=ex, prec_search=ps)
        o = po.oracle_encode(pcm, bps, rate, level, exhaustive=ex, prec_search=ps)
        assert o["data"] == _frames(r), (fam, bps, level, mode)


@pytest.mark.parametrize("order", [1, 2, 4, 5, 7, 9, 10, 11, 13, 14, 15, 16, 17, 24, 31, 32])
def test_every_lpc_order_class(ref, order):
    """max_lpc_order selects the compiled autocorrelation routine (lag 8 / 12 / 16, stream_encoder.c:1058-1066) and the
    FIR width: one value from every class, 16-bit at 96 kHz (orders above 12 are not in the subset at <= 48 kHz) and 24-bit"""
    for bps in (16, 24):
        pcm = signals.music(4096 * 2 + 501, 2, bps, seed=order)
        for level in (5, 8):
            r = po.ref_encode(pcm, bps, 96000, level, max_lpc_order=order)
            o = po.oracle_encode(pcm, bps, 96000, level, max_lpc_order=order)
            assert o["data"] == _frames(r), (order, bps, level)


@pytest.mark.parametrize("blocksize", [192, 256, 576, 1000, 1024, 2304, 4000, 4608, 8192, 16384])
def test_block_sizes(ref, blocksize):
    for bps, level in ((16, 2), (16, 8), (24, 8)):
        pcm = signals.music(blocksize * 3 + blocksize // 3 + 7, 2, bps, seed=blocksize % 97)
        r = po.ref_encode(pcm, bps, 96000, level, blocksize=blocksize)
        assert po.oracle_encode(pcm, bps, 96000, level, blocksize=blocksize)["data"] == _frames(r), (blocksize, bps, level)


@pytest.mark.parametrize("po_range", [(0, 0), (0, 2), (3, 3), (2, 6), (0, 8), (8, 8)])
def test_partition_order_ranges(ref, po_range):
    lo, hi = po_range
    pcm = signals.mixed(4096 * 3 + 99, 2, 16)
    for level in (2, 8):
        r = po.ref_encode(pcm, 16, 44100, level, min_po=lo, max_po=hi)
        assert po.oracle_encode(pcm, 16, 44100, level, min_po=lo, max_po=hi)["data"] == _frames(r), (po_range, level)


TINY_ORDERS = (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32)
DISABLES = ((0, 0, 0), (1, 0, 1), (1, 1, 1))


@pytest.mark.parametrize("blocksize", range(16, 34))
def test_tiny_blocks_all_orders(ref, blocksize):
    """test/test_streams.sh:221-239: 8-bit mono noise, -8 -p -e -l <order> --lax --blocksize=<16..33>, with the
    constant / fixed / verbatim subframes switched off in turn, and the same with subdivide_tukey(32)"""
    pcm = signals.white(blocksize * 5 + 7, 1, 8, seed=blocksize)
    for order in TINY_ORDERS:
        if order > blocksize:
            continue
        for kw in [dict(disable=d) for d in DISABLES] + [dict(apodization="subdivide_tukey(32)")]:
            okw = dict(kw)
            if "apodization" in okw:
                okw["apod"] = ("subdivide_tukey", 32)
                del okw["apodization"]
            r = po.ref_encode(pcm, 8, 44100, 8, blocksize=blocksize, max_lpc_order=order, exhaustive=1, prec_search=1,
                              streamable_subset=0, **kw)
            o = po.oracle_encode(pcm, 8, 44100, 8, blocksize=blocksize, max_lpc_order=order, exhaustive=1, prec_search=1, **okw)
            assert o["data"] == _frames(r), (blocksize, order, kw)


@pytest.mark.parametrize("order", [16, 32])
def test_high_orders_with_searches(ref, order):
    """test/test_streams.sh:181-219: -0 -l 16|32 --lax -m -e -p on sines and full-scale streams, 8 / 16 / 24 bits"""
    for bps in (8, 16, 24):
        for fam, ch in (("sine", 1), ("music", 2), ("square", 2)):
            pcm = signals.FAMILIES[fam](1152 * 2 + 301, ch, bps)
            kw = dict(max_lpc_order=order, exhaustive=1, prec_search=1, mid_side=1, loose_mid_side=0)
            r = po.ref_encode(pcm, bps, 44100, 0, streamable_subset=0, **kw)
            kw["loose"] = kw.pop("loose_mid_side")
            o = po.oracle_encode(pcm, bps, 44100, 0, **kw)
            assert o["data"] == _frames(r), (order, bps, fam)


@pytest.mark.parametrize("kw", [dict(blocksize=1000), dict(blocksize=4096, max_lpc_order=20), dict(blocksize=576),
                                dict(blocksize=33, max_lpc_order=32)], ids=["b1000", "l20", "b576", "b33l32"])
def test_constant_and_silent_channels(ref, kw):
    for fam in ("square", "constant", "silence", "mixed"):
        pcm = signals.FAMILIES[fam](kw["blocksize"] * 3 + 77, 2, 16)
        r = po.ref_encode(pcm, 16, 44100, 8, streamable_subset=0, **kw)
        o = po.oracle_encode(pcm, 16, 44100, 8, **kw)
        assert o["data"] == _frames(r), (fam, kw)


@pytest.mark.parametrize("rate", [9, 90, 8000, 22050, 90000, 96000, 192000, 352800, 655350, 1048575])
def test_sample_rate_codes(ref, rate):
    """test/test_streams.sh:241-250 frame-header variations: the sample-rate field (framing.c:289-329)"""
    pcm = signals.music(4096 + 300, 1, 16, seed=rate % 97)
    r = po.ref_encode(pcm, 16, rate, 5, streamable_subset=0)
    o = po.oracle_encode(pcm, 16, rate, 5)
    assert o["data"] == _frames(r), rate


@pytest.mark.parametrize("bps", [25, 27, 28, 31, 32])
def test_wide_samples(ref, bps):
    """more than 24 bits per sample: the overflow-checked estimators and residuals (fixed_intrin_avx2.c:187, fixed.c:424,
    lpc.c:832,886), 64-bit fixed residuals, and at 32 bits the 33-bit side channel (stream_encoder.c:3831-3835,5103)"""
    for fam in ("music", "white", "sine", "square", "mixed", "wasted", "quiet", "constant", "silence"):
        for level in (0, 1, 2, 5, 8):
            for ch in (1, 2):
                pcm = signals.FAMILIES[fam](4096 + 1333, ch, bps)
                r = po.ref_encode(pcm, bps, 96000, level, streamable_subset=0)
                o = po.oracle_encode(pcm, bps, 96000, level)
                assert o["data"] == _frames(r), (bps, fam, level, ch)


@pytest.mark.parametrize("bps", [8, 16, 24, 32])
@pytest.mark.parametrize("pattern", range(1, 8))
def test_full_scale_deflection(ref, bps, pattern):
    """test/test_streams.sh:188-193: fsd<bps>-0<pattern>, -0 -l 16 --lax -m -e -p; mono like the suite, and as an inverted pair"""
    for ch in (1, 2):
        pcm = signals.fsd(1152 * 2 + 100, ch, bps, pattern)
        kw = dict(max_lpc_order=16, exhaustive=1, prec_search=1, mid_side=1)
        r = po.ref_encode(pcm, bps, 44100, 0, streamable_subset=0, loose_mid_side=0, **kw)
        o = po.oracle_encode(pcm, bps, 44100, 0, loose=0, **kw)
        assert o["data"] == _frames(r), (bps, pattern, ch)


@pytest.mark.parametrize("tail", [1, 5, 6, 7, 9, 33, 1001, 4095])
def test_wide_samples_short_last_block(ref, tail):
    """(n-4) % 4 != 0 in the four-lane estimators of the overflow-checked flavour"""
    for bps in (28, 32):
        for fam in ("music", "white"):
            pcm = signals.FAMILIES[fam](4096 + tail, 2, bps)
            for level, kw in ((5, {}), (8, dict(exhaustive=1))):
                r = po.ref_encode(pcm, bps, 96000, level, streamable_subset=0, **kw)
                o = po.oracle_encode(pcm, bps, 96000, level, **kw)
                assert o["data"] == _frames(r), (bps, fam, tail, level)


@pytest.mark.parametrize("kind", range(4))
def test_overflow_checked_residual_at_24_bits(ref, kind):
    """smooth anti-phase pairs: candidates whose bound on the residual width exceeds 32 bits (lpc.c:962) take
    FLAC__lpc_compute_residual_from_qlp_coefficients_limit_residual (lpc.c:832) already at 24 bits per sample"""
    for bps in (24, 20):
        pcm = signals.slow(4096 * 2 + 77, 2, bps, kind)
        for level, kw in ((5, {}), (8, {}), (8, dict(exhaustive=1)), (5, dict(prec_search=1))):
            r = po.ref_encode(pcm, bps, 96000, level, streamable_subset=0, **kw)
            o = po.oracle_encode(pcm, bps, 96000, level, **kw)
            assert o["data"] == _frames(r), (kind, bps, level, kw)


ALL_WINDOW_SPECS = [("bartlett", "bartlett", ()), ("bartlett_hann", "bartlett_hann", ()), ("blackman", "blackman", ()),
                    ("blackman_harris_4term_92db", "blackman_harris_4term_92db_sidelobe", ()), ("connes", "connes", ()),
                    ("flattop", "flattop", ()), ("gauss(0.3)", "gauss", (0.3,)), ("gauss(0.05)", "gauss", (0.05,)), ("hamming", "hamming", ()),
                    ("hann", "hann", ()), ("kaiser_bessel", "kaiser_bessel", ()), ("nuttall", "nuttall", ()), ("rectangle", "rectangle", ()),
                    ("triangle", "triangle", ()), ("tukey(0.5)", "tukey", (0.5,)), ("tukey(0.03)", "tukey", (0.03,)), ("welch", "welch", ()),
                    ("partial_tukey(3/0.3/0.5)", None, ()), ("punchout_tukey(2/0.2/0.4)", None, ())]


@pytest.mark.parametrize("spec,fn,args", ALL_WINDOW_SPECS, ids=[s[0] for s in ALL_WINDOW_SPECS])
def test_host_window_tables_bit_exact(ref, spec, fn, args):
    """flac_amd/csrc/host/window.c against the reference's compiled window.c, table by table and bit for bit (the
    reference's build regroups the cosine sums: -fassociative-math)"""
    lib = po.load_ref()
    for L in (4096, 2206, 1152, 333, 33, 17, 16):
        s = engine.make_settings(2, 16, 44100, 5, apodization=spec, blocksize=4096)
        w = engine.host_windows(s, L)
        if fn is not None:
            want = np.zeros(L, dtype=np.float32)
            f = getattr(lib, "FLAC__window_" + fn)
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_int32] + [C.c_float] * len(args)
            f(want.ctypes.data, L, *[C.c_float(a) for a in args])
            assert np.array_equal(w[0].view(np.uint32), want.view(np.uint32)), (spec, L)
        else:
            # the multi-window specs expand into several partial / punch-out tukey windows (stream_encoder.c:1973-2030)
            kind, rest = spec.split("(")
            n, p_, _ = rest[:-1].split("/")[0], *rest[:-1].split("/")[1:]
            n = int(n)
            tukey_p = float(rest[:-1].split("/")[2]) if len(rest[:-1].split("/")) > 2 else 0.5
            overlap = float(rest[:-1].split("/")[1])
            overlap_units = 1.0 / (1.0 - overlap) - 1.0
            f = getattr(lib, "FLAC__window_" + kind)
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float]
            k = 0
            for m in range(n):
                start = np.float32(m / (n + overlap_units))
                end = np.float32((m + 1 + overlap_units) / (n + overlap_units))
                if kind == "punchout_tukey":
                    pass
                want = np.zeros(L, dtype=np.float32)
                f(want.ctypes.data, L, C.c_float(tukey_p), C.c_float(float(start)), C.c_float(float(end)))
                assert np.array_equal(w[k].view(np.uint32), want.view(np.uint32)), (spec, L, m)
                k += 1


@pytest.mark.parametrize("blocksize", [16385, 20000, 32768, 65535])
def test_blocks_longer_than_16384(ref, blocksize):
    """test/test_streams.sh:243 (-b 65535)"""
    for fam, ch, bps, level, kw in (("music", 2, 16, 5, {}), ("sine", 1, 16, 0, dict(max_lpc_order=32, exhaustive=1)),
                                    ("mixed", 2, 24, 8, {}), ("white", 2, 32, 5, {}), ("wasted", 3, 16, 2, {})):
        pcm = signals.FAMILIES[fam](blocksize + 777, ch, bps)
        r = po.ref_encode(pcm, bps, 44100, level, blocksize=blocksize, streamable_subset=0, **kw)
        o = po.oracle_encode(pcm, bps, 44100, level, blocksize=blocksize, **kw)
        assert o["data"] == _frames(r), (blocksize, fam, ch, bps, level)


@pytest.mark.parametrize("ch", range(1, 9))
def test_every_channel_count(ref, ch):
    """1..8 channels (the frame header's channel assignment 0..7, stream_encoder_framing.c:245-391; the seeded sweeps draw from
    1, 2, 3, 6, 8): independent channels whatever the preset asks of stereo, limit_min_bitrate's last-channel rule
    (stream_encoder.c:3874-3879) with every other channel constant, 16- and 24-bit, a short last block"""
    for bps, rate in ((16, 48000), (24, 96000)):
        base = signals.music(4096 * 2 + 333, ch, bps, seed=10 * ch + bps)
        flat = base.copy()
        flat[:, :] = 5
        flat[:, -1] = base[:, -1] if ch > 1 else 5
        one_const = base.copy()
        one_const[:, ch // 2] = -7
        for level, pcm, kw in ((8, base, {}), (5, base, {}), (0, base, dict(blocksize=4096)), (8, flat, dict(limit_min_bitrate=1)),
                               (5, one_const, dict(limit_min_bitrate=1)), (8, np.zeros_like(base), dict(limit_min_bitrate=1))):
            r = po.ref_encode(pcm, bps, rate, level, **kw)
            assert po.oracle_encode(pcm, bps, rate, level, **kw)["data"] == _frames(r), (ch, bps, level, kw)
