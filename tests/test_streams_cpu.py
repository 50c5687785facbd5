"""CPU: the streams of the reference's own generator (src/test_streams/main.c, built into oracle/_ref/test_streams) under the
options its test script encodes them with (test/test_streams.sh:178-219, BASELINE.json config 1): oracle == reference."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

GEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "test_streams")
pytestmark = pytest.mark.skipif(not (os.path.exists(GEN) and po.have_ref()), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def streams(tmp_path_factory):
    d = tmp_path_factory.mktemp("streams")
    subprocess.run([GEN], cwd=str(d), check=True, capture_output=True, timeout=300)
    return str(d)


def _raw(path, channels, bps):
    """--force-raw-format --endian=little --sign=signed (8-bit: signed bytes), as test_streams.sh feeds them"""
    b = np.fromfile(path, dtype=np.uint8)
    w = bps // 8
    b = b[: len(b) // (w * channels) * w * channels].reshape(-1, w)
    v = np.zeros(len(b), dtype=np.int64)
    for k in range(w):
        v |= b[:, k].astype(np.int64) << (8 * k)
    v = np.where(v >= 1 << (bps - 1), v - (1 << bps), v)
    return v.astype(np.int32).reshape(-1, channels)


def _check(pcm, bps, rate, level, **kw):
    rkw = dict(kw)
    okw = dict(kw)
    if "loose_mid_side" in okw:
        okw["loose"] = okw.pop("loose_mid_side")
    r = po.ref_encode(pcm, bps, rate, level, streamable_subset=0, **rkw)
    o = po.oracle_encode(pcm, bps, rate, level, **okw)
    assert o["data"] == r["data"][r["header_bytes"]:]


def test_config_1_sines_at_5(streams):
    """BASELINE.json configs[0]: flac -5 on sine16-02, -03, -04 (200 000 mono samples each, 49 frames)"""
    for nn in ("02", "03", "04"):
        pcm = _raw(os.path.join(streams, "sine16-%s.raw" % nn), 1, 16)
        assert len(pcm) == 200000
        r = po.ref_encode(pcm, 16, 44100, 5)
        assert len(r["frame_bytes"]) == 49
        o = po.oracle_encode(pcm, 16, 44100, 5)
        assert o["data"] == r["data"][r["header_bytes"]:]


@pytest.mark.parametrize("bps", [8, 16, 24, 32])
def test_full_scale_deflection_and_sines(streams, bps):
    """fsd<bps>-01..07 and sine<bps>-00..19 with "-0 -l 16 --lax -m -e [-p]" (test_streams.sh:186-219); the long sines are cut to
    20 000 samples per channel to keep the CPU suite short"""
    for nn in range(1, 8):
        pcm = _raw(os.path.join(streams, "fsd%d-%02d.raw" % (bps, nn)), 1, bps)
        _check(pcm, bps, 44100, 0, max_lpc_order=16, exhaustive=1, prec_search=1)
    for nn, ch, rate in ((0, 1, 48000), (1, 1, 96000), (2, 1, 44100), (3, 1, 44100), (4, 1, 44100), (10, 2, 48000), (11, 2, 48000),
                         (12, 2, 96000), (13, 2, 44100), (15, 2, 44100), (17, 2, 44100), (19, 2, 44100)):
        pcm = _raw(os.path.join(streams, "sine%d-%02d.raw" % (bps, nn)), ch, bps)[:20000]
        kw = dict(max_lpc_order=16, exhaustive=1)
        if ch == 2:
            kw.update(mid_side=1, loose_mid_side=0)
        _check(pcm, bps, rate, 0, **kw)


def test_wasted_bits_small_files_and_noise(streams):
    _check(_raw(os.path.join(streams, "wbps16-01.raw"), 1, 16), 16, 44100, 0, max_lpc_order=16, exhaustive=1, prec_search=1)
    for nn, ch in (("01", 1), ("02", 2), ("03", 1), ("04", 2)):
        pcm = _raw(os.path.join(streams, "test%s.raw" % nn), ch, 16)
        kw = dict(max_lpc_order=16, exhaustive=1, prec_search=1)
        if ch == 2:
            kw.update(mid_side=1, loose_mid_side=0)
        if len(pcm):
            _check(pcm, 16, 44100, 0, **kw)
    pcm = _raw(os.path.join(streams, "noise.raw"), 1, 8)[:60000]
    _check(pcm, 8, 44100, 0)
    pcm = _raw(os.path.join(streams, "noise.raw"), 2, 16)[:60000]
    for level in (1, 5, 8):
        _check(pcm, 16, 44100, level)
