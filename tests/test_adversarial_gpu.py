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
