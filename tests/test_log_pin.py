"""Pins the libm dependence of the model search: the two expressions that go through log()
(lpc.c:1594, fixed.c:284-288) are evaluated by the engine with flac_amd/csrc/flacgpu_log.h -- glibc 2.35's algorithm
restated operation by operation -- and must equal the libm the reference binary links, bit for bit.

CPU: the restatement compiled for the host (oracle/liblogpin.so) against this box's libm on 1.2e8 arguments.
GPU: the device instantiation (flacgpu_debug_log_kat) against the same libm on 2.4e7 arguments over the reachable
domain (err*0.5/N over 2^-60..2^70; err*ln2/n for integer error sums), plus the two folded expressions themselves."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN_SO = os.path.join(ROOT, "oracle", "liblogpin.so")


def _pin():
    if not os.path.exists(PIN_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    lib = C.CDLL(PIN_SO)
    lib.logpin_compare.restype = C.c_size_t
    lib.logpin_compare.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_double)]
    lib.logpin_libm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.logpin_expected_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.logpin_fixed_rbps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def _domain(kind, n, rng):
    if kind == 0:      # wide range of magnitudes
        return np.exp(rng.uniform(-80, 140, n))
    if kind == 1:      # the separate polynomial near 1
        return rng.uniform(0.9, 1.1, n)
    if kind == 2:      # every binade incl. subnormals
        return np.ldexp(rng.uniform(1, 2, n), rng.integers(-1070, 1020, n))
    # fixed.c:284: integer error sums times ln2 over block lengths
    return rng.integers(1, 2 ** 50, n).astype(np.float64) * 0.6931471805599453 / rng.integers(12, 65531, n)


SPECIAL = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 5e-324, 2.2250738585072014e-308, 1.0, np.nextafter(1, 0), np.nextafter(1, 2),
                    0.9375, np.nextafter(0.9375, 0), 1.064697265625, np.nextafter(1.064697265625, 2), np.nextafter(1.064697265625, 0),
                    1.7976931348623157e308, 0.5, 2.0, 0.6931471805599453])


def test_restated_log_equals_this_boxs_libm():
    lib = _pin()
    rng = np.random.default_rng(20260922)
    total = 0
    for trial in range(12):
        x = _domain(trial % 4, 10_000_000, rng)
        fb = C.c_double(0)
        bad = lib.logpin_compare(x.ctypes.data, x.size, None, C.byref(fb))
        assert bad == 0, "restated log differs from libm on %d of %d arguments, first %r" % (bad, x.size, fb.value)
        total += x.size
    fb = C.c_double(0)
    assert lib.logpin_compare(SPECIAL.ctypes.data, SPECIAL.size, None, C.byref(fb)) == 0, fb.value
    assert total >= 10 ** 8


@pytest.mark.gpu
def test_device_log_equals_libm():
    from flac_amd import engine
    eng = engine.load_engine()
    eng.flacgpu_debug_log_kat.restype = C.c_int
    eng.flacgpu_debug_log_kat.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib = _pin()
    rng = np.random.default_rng(7)
    ocml_diff = 0
    for kind in range(4):
        x = np.concatenate([_domain(kind, 6_000_000, rng), SPECIAL])
        want = np.empty_like(x)
        lib.logpin_libm(x.ctypes.data, x.size, want.ctypes.data)
        got = np.empty_like(x)
        assert eng.flacgpu_debug_log_kat(0, 0, x.ctypes.data, None, x.size, got.ctypes.data) == 0
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "device log != libm at %r" % x[~same][:4]
        dev = np.empty_like(x)
        assert eng.flacgpu_debug_log_kat(0, 3, x.ctypes.data, None, x.size, dev.ctypes.data) == 0
        ocml_diff += int((dev.view(np.uint64) != want.view(np.uint64)).sum())
    print("arguments on which the device library's own log differs from libm: %d of %d" % (ocml_diff, 4 * (6_000_000 + SPECIAL.size)))


@pytest.mark.gpu
def test_device_expected_bits_and_fixed_estimate_equal_the_compiled_reference_expressions():
    from flac_amd import engine
    eng = engine.load_engine()
    eng.flacgpu_debug_log_kat.restype = C.c_int
    eng.flacgpu_debug_log_kat.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib = _pin()
    rng = np.random.default_rng(11)
    n = 6_000_000
    # lpc.c:1591-1606: lpc_error from Levinson-Durbin (any positive magnitude, some zero / negative), error_scale = 0.5/N
    err = np.exp(rng.uniform(-40, 120, n))
    err[::1000] = 0.0
    err[1::1000] = -err[1::1000]
    scale = 0.5 / rng.integers(16, 65536, n).astype(np.float64)
    want = np.empty(n)
    lib.logpin_expected_bits(err.ctypes.data, scale.ctypes.data, n, want.ctypes.data)
    got = np.empty(n)
    assert eng.flacgpu_debug_log_kat(0, 1, err.ctypes.data, scale.ctypes.data, n, got.ctypes.data) == 0
    assert (got.view(np.uint64) == want.view(np.uint64)).all()
    # fixed.c:284-288: total_error (uint64 below 2^53 here, exact in double) and data_len
    e = rng.integers(0, 2 ** 50, n).astype(np.uint64)
    e[::7] = rng.integers(0, 4096, e[::7].size).astype(np.uint64)
    n4 = rng.integers(12, 65531, n).astype(np.uint32)
    wantf = np.empty(n, dtype=np.float32)
    lib.logpin_fixed_rbps(e.ctypes.data, n4.ctypes.data, n, wantf.ctypes.data)
    gotd = np.empty(n)
    ed, nd = e.astype(np.float64), n4.astype(np.float64)
    assert eng.flacgpu_debug_log_kat(0, 2, ed.ctypes.data, nd.ctypes.data, n, gotd.ctypes.data) == 0
    assert (gotd.astype(np.float32).view(np.uint32) == wantf.view(np.uint32)).all()
