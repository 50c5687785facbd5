"""-m gpu: the HIP path against the oracle on the adversarial signals of tests/test_adversarial_cpu.py (where the oracle is held to the
real reference on the same cases): resonances on the unit circle, exact polynomials, impulses, full-scale alternation and noise at
every width, random walks, clusters of close tones, copies / negations / shifted copies as the other channel, the widest wasted bits --
under the random configurations of the seeded sweep, device verify on, scratch memory poisoned."""
import os

import numpy as np
import pytest

from test_adversarial_cpu import adversarial_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("FLACGPU_ADV_SEEDS", "40"))))
def test_adversarial_signals_gpu_vs_oracle(seed, monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    for sub in range(8):
        pcm, ch, bps, rate, kw, s = adversarial_case(seed * 8 + sub)
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
        try:
            eng.set_verify(True)
            data, fb = eng.encode(pcm)
            v = eng.last_verify_result()
            assert v.status == 0, ("verify", v.status, v.frame_number, v.channel, v.sample, seed, sub, ch, bps, rate, kw)
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (seed, sub, ch, bps, rate, kw)


def test_fixed_sums_at_the_32_bit_edge_gpu_vs_oracle(monkeypatch):
    """prep3_kernel / prep4_kernel summed a quarter's fourth differences in 32 bits up to 20-bit input; from 18 (17 with a
    side channel) bits on that wraps on a Nyquist alternation near half of full scale and the wrong fixed order is guessed.  The same
    cases as the CPU test that holds the oracle to the reference (tests/test_adversarial_cpu.py::edge_sum_cases)."""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    from test_adversarial_cpu import edge_sum_cases
    monkeypatch.setenv("FLACGPU_POISON", "1")
    for name, pcm, ch, bps, kw in edge_sum_cases():
        for level in (2, 5):
            s = flac_amd.make_settings(ch, bps, 48000, level, **kw)
            eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
            try:
                data, fb = eng.encode(pcm)
            finally:
                eng.close()
            o = oracle_encode_settings(pcm, s)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, level)
