"""CPU: the seeded configuration sweep of test_gpu_parity.py::test_random_configurations, oracle (driven by the host layer's
resolved settings and window tables) against the real reference."""
import os

import numpy as np
import pytest

import signals
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FLACGPU_TEST_SEEDS", "40"))))
def test_random_configurations_oracle_vs_reference(seed):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    from test_gpu_parity import _random_config
    rng = np.random.default_rng(1000 + seed)
    done = 0
    for _ in range(14):
        fam, n, ch, bps, rate, kw = _random_config(rng)
        pcm = signals.FAMILIES[fam](n, ch, bps)
        try:
            s = flac_amd.make_settings(ch, bps, rate, 5, **kw)
        except flac_amd.FlacGpuError:
            continue
        rkw = dict(blocksize=kw["blocksize"], max_lpc_order=kw["max_lpc_order"], streamable_subset=0, min_po=kw["min_partition_order"],
                   max_po=kw["max_partition_order"], limit_min_bitrate=kw["limit_min_bitrate"], disable=kw["disable"],
                   exhaustive=kw.get("exhaustive", 0), prec_search=kw.get("prec_search", 0))
        if "mid_side" in kw:
            rkw["mid_side"], rkw["loose_mid_side"] = kw["mid_side"], kw["loose_mid_side"]
        if "qlp_coeff_precision" in kw:
            rkw["qlp_precision"] = kw["qlp_coeff_precision"]
        if "apodization" in kw:
            rkw["apodization"] = kw["apodization"]
        try:
            r = po.ref_encode(pcm, bps, rate, 5, **rkw)
        except RuntimeError:
            continue                               # the reference itself gives up (every subframe type disabled on a constant signal)
        o = oracle_encode_settings(pcm, s)
        assert o["data"] == r["data"][r["header_bytes"]:], (fam, n, ch, bps, rate, kw)
        done += 1
    assert done >= 6


def test_many_apodizations_and_deep_subdivision_oracle_vs_reference():
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    specs = [";".join(["hann", "welch", "tukey(0.3)", "gauss(0.2)", "blackman", "flattop", "nuttall", "bartlett"] * 4),
             "subdivide_tukey(12)", "subdivide_tukey(32)", "tukey(0.5);partial_tukey(4);punchout_tukey(5);subdivide_tukey(7)"]
    for spec in specs:
        for ch, bps, n in ((2, 16, 4096 + 321), (1, 24, 4096 * 2)):
            pcm = signals.music(n, ch, bps, seed=len(spec))
            s = flac_amd.make_settings(ch, bps, 44100, 8, apodization=spec, streamable_subset=0)
            r = po.ref_encode(pcm, bps, 44100, 8, apodization=spec, streamable_subset=0)
            o = oracle_encode_settings(pcm, s)
            assert o["data"] == r["data"][r["header_bytes"]:], (spec, ch, bps)
