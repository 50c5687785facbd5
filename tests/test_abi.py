"""The C-ABI libraries load and export every symbol include/flacgpu.h declares; without a GPU the
engine refuses loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from flac_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flacgpu_[a-z0-9_]+)\s*\(", text)))


def test_engine_exports_every_declared_symbol():
    names = _declared_functions(os.path.join(ROOT, "include", "flacgpu.h"))
    assert "flacgpu_create" in names and "flacgpu_encode_batch" in names and "flacgpu_encode_batch_device" in names
    lib = C.CDLL(engine.ENGINE_SO)
    for n in names:
        assert hasattr(lib, n), "libflacgpu.so does not export %s" % n


def test_host_layer_exports():
    lib = engine.load_host()
    for n in ("flacgpu_host_settings_defaults", "flacgpu_host_settings_level", "flacgpu_host_settings_apodization",
              "flacgpu_host_settings_resolve", "flacgpu_host_engine_config", "flacgpu_host_windows",
              "flacgpu_host_md5_init", "flacgpu_host_md5_update", "flacgpu_host_md5_final", "flacgpu_host_md5_pcm"):
        assert hasattr(lib, n), n


def test_settings_resolution_matches_reference_defaults():
    # stream_encoder.c:748-795: default blocksize and qlp precision
    for level, bs, prec in ((0, 1152, 10), (2, 1152, 10), (3, 4096, 12), (5, 4096, 12), (8, 4096, 12)):
        s = engine.make_settings(2, 16, 44100, level)
        assert (s.blocksize, s.qlp_coeff_precision) == (bs, prec)
    assert engine.make_settings(2, 24, 96000, 8).qlp_coeff_precision == 15
    assert engine.make_settings(2, 8, 44100, 8).qlp_coeff_precision == 6
    s = engine.make_settings(1, 16, 44100, 8)
    assert s.do_mid_side_stereo == 0            # mid/side only for stereo (:737)
    s = engine.make_settings(2, 16, 44100, 8)
    assert (s.apodizations[0].type, s.apodizations[0].parts) == (16, 3)
    with pytest.raises(engine.FlacGpuError):    # not streamable subset: NOT_STREAMABLE status
        engine.make_settings(2, 16, 44100, 8, blocksize=8192)


def test_every_apodization_the_reference_takes_reaches_the_engine_config():
    s = engine.make_settings(2, 16, 44100, 8, apodization=";".join(["hann", "welch"] * 16))      # FLAC__MAX_APODIZATION_FUNCTIONS = 32
    cfg = engine.EngineConfig()
    assert s.num_apodizations == 32
    assert engine.load_host().flacgpu_host_engine_config(C.byref(s), 0, 16, C.byref(cfg)) == 0
    assert cfg.num_apodizations == 32


def test_unsupported_configuration_is_refused_not_emulated():
    """the engine itself refuses what it cannot do (there is no CPU path to fall back to)"""
    s = engine.make_settings(2, 16, 44100, 8)
    cfg = engine.EngineConfig()
    assert engine.load_host().flacgpu_host_engine_config(C.byref(s), 0, 16, C.byref(cfg)) == 0
    cfg.bits_per_sample = 33
    ctx = C.c_void_p()
    w = engine.host_windows(s, s.blocksize)
    lib = engine.load_engine()
    lib.flacgpu_create.restype = C.c_int
    assert lib.flacgpu_create(C.byref(cfg), w.ctypes.data, C.byref(ctx)) == -1      # FLACGPU_ERR_UNSUPPORTED, before any device is touched


def test_wider_searches_reach_the_engine_config():
    s = engine.make_settings(2, 16, 44100, 8, exhaustive=1, prec_search=1)
    cfg = engine.EngineConfig()
    assert engine.load_host().flacgpu_host_engine_config(C.byref(s), 0, 16, C.byref(cfg)) == 0
    assert (cfg.abi_version, cfg.do_exhaustive_model_search, cfg.do_qlp_coeff_prec_search) == (5, 1, 1)


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.FlacGpuError, match="no usable HIP device"):
        engine.FrameEngine(engine.make_settings(2, 16, 44100, 8))
