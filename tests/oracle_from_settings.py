"""Drive the oracle with the HOST layer's resolved settings and window tables (flac_amd/csrc/host/settings.c, window.c),
so that any configuration make_settings() accepts can be checked GPU-vs-oracle.  The host tables themselves are pinned
against the reference in test_oracle_vs_ref.py::test_host_window_tables_and_apodization_parser."""
import ctypes as C

import numpy as np

from flac_amd import engine
from oracle import pyoracle as po

SUBDIVIDE_TUKEY = 16      # FGH_APOD_SUBDIVIDE_TUKEY (flacgpu_host.h), FLAC__APODIZATION_SUBDIVIDE_TUKEY in the reference


def _frame_config(s, blk):
    c = po.FoConfig()
    c.channels, c.bits_per_sample, c.sample_rate, c.blocksize = s.channels, s.bits_per_sample, s.sample_rate, blk
    ms = int(bool(s.do_mid_side_stereo) and s.channels == 2)
    c.do_mid_side, c.loose_mid_side = ms, int(bool(s.loose_mid_side_stereo) and ms)
    c.max_lpc_order, c.qlp_coeff_precision = s.max_lpc_order, s.qlp_coeff_precision
    c.min_partition_order, c.max_partition_order = s.min_residual_partition_order, s.max_residual_partition_order
    c.num_apodizations = s.num_apodizations if s.max_lpc_order else 0
    keep = engine.host_windows(s, blk) if c.num_apodizations else None
    for a in range(c.num_apodizations):
        sub = s.apodizations[a].type == SUBDIVIDE_TUKEY
        c.apodizations[a].kind = 1 if sub else 0
        c.apodizations[a].parts = s.apodizations[a].parts if sub else 0
        c.apodizations[a].window = keep[a].ctypes.data_as(C.POINTER(C.c_float))
    lpc = s.max_lpc_order
    c.autoc_variant = 8 if lpc < 8 else 12 if lpc < 12 else 16 if lpc < 16 else 0
    c.disable_constant, c.disable_fixed, c.disable_verbatim = s.disable_constant_subframes, s.disable_fixed_subframes, s.disable_verbatim_subframes
    c.limit_min_bitrate = s.limit_min_bitrate
    c.exhaustive, c.prec_search = int(bool(s.do_exhaustive_model_search)), int(bool(s.do_qlp_coeff_prec_search) and lpc > 0)
    return c, keep


def oracle_encode_settings(pcm, s, first_frame=0):
    """pcm int32 [n, channels], s a resolved HostSettings -> dict(data, frame_bytes): the audio frames"""
    lib = po.load_oracle()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    N = s.blocksize
    out = np.empty(N * ch * 5 + 65536, dtype=np.uint8)
    frames, fb = [], []
    cfgs = {}
    pos, fn = 0, first_frame
    while pos < n:
        blk = min(N, n - pos)
        if blk not in cfgs:
            cfgs[blk] = _frame_config(s, blk)
        planar = np.ascontiguousarray(pcm[pos:pos + blk].T)
        ptrs = (C.c_void_p * ch)(*[planar[i].ctypes.data for i in range(ch)])
        r = lib.fo_encode_frame(C.byref(cfgs[blk][0]), ptrs, fn, out.ctypes.data, out.size, None)
        if r < 0:
            raise RuntimeError("oracle encode failed: %d" % r)
        frames.append(out[:r].tobytes())
        fb.append(r)
        pos += blk
        fn += 1
    return dict(data=b"".join(frames), frame_bytes=np.array(fb, dtype=np.uint32))
