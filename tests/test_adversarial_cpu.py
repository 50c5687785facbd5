"""CPU: oracle against the real reference on signals built to reach the corners of the arithmetic rather than to sound like music --
test infrastructure pinning test infrastructure: the oracle is what the GPU path is held to, so it is held to the reference on more
than the nine signal families of the sweep (tests/test_random_cpu.py).  Seeded: FLACGPU_ADV_SEEDS (default 24; 2000 seeds = 16 000 cases ran
clean when this file was written, and 1000 of them on the GPU: profiles/archive/r05_soak_seeded_sweep.txt).

What the generators aim at (reference file:line):
  * resonances on and next to the unit circle, growing and decaying exponentials -- coefficients of 2^k and more: the negative-shift
    branch of FLAC__lpc_quantize_coefficients (lpc.c:283-313), the clamp to [-2^(p-1), 2^(p-1)-1] (:262-276), precision limits
    (stream_encoder.c:4591-4595)
  * ramps, parabolas, cubics: a fixed predictor's error is exactly 0 (fixed.c:284-300: log of 0 guarded, rbps 0, the CONSTANT test of
    stream_encoder.c:4111) and Levinson-Durbin stops early with err == 0 (lpc.c:200-217)
  * impulses, sparse bursts, a block that is silent but for its warm-up samples: partition sums of 0, Rice parameter 0, escape-free
    partitions of one sample's energy (stream_encoder.c:4954-5075)
  * full-scale alternation and full-scale noise at every width: 32-bit wrap of the residual, the overflow-checked FIR (lpc.c:832,886),
    the 64-bit fixed predictors (fixed.c:301-424), verbatim as the winner
  * random walks (a 1/f^2 spectrum), sums of a dozen close tones (ill-conditioned autocorrelation matrices), a tone at Nyquist
  * channels that are copies, negations, shifted copies, one constant -- the mid/side decision's ties (stream_encoder.c:3955-3971)
  * per-channel wasted bits incl. the widest (samples that are 0 or +-2^(bps-1)), and offsets that make every sample odd."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built")


def _fit(x, bps):
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    return np.clip(np.rint(x), lo, hi).astype(np.int64)


def adversarial_channel(rng, n, bps):
    fs = float(1 << (bps - 1))
    t = np.arange(n, dtype=np.float64)
    kind = int(rng.integers(0, 16))
    if kind == 0:            # a resonance at radius r (<, =, > 1) scaled to end (or start) near full scale
        r = float(rng.choice([0.9, 0.99, 0.999, 1.0, 1.0, 1.001, 1.01]))
        w = float(rng.uniform(0.001, np.pi))
        e = r ** t if r <= 1.0 else r ** (t - n)
        x = fs * float(rng.uniform(0.05, 0.99)) * e * np.cos(w * t + rng.uniform(0, 6.28))
    elif kind == 1:          # polynomial of degree 0..4 (exact for the fixed predictors)
        deg = int(rng.integers(0, 5))
        u = t / max(n - 1, 1) - float(rng.uniform(0, 1))
        x = fs * float(rng.uniform(0.01, 0.9)) * u ** deg * (1 if rng.random() < 0.5 else -1)
        if rng.random() < 0.5:
            x = np.rint(x / 7.0) * 7.0
    elif kind == 2:          # impulses
        x = np.zeros(n)
        k = int(rng.integers(1, 6))
        x[rng.integers(0, n, size=k)] = rng.uniform(-fs, fs, size=k)
    elif kind == 3:          # silent but for the first few samples of each 64
        x = np.zeros(n)
        m = int(rng.integers(1, 33))
        period = int(rng.choice([64, 576, 1152, 4096, n + 1]))
        idx = np.arange(n) % period < m
        x[idx] = rng.uniform(-fs, fs, size=int(idx.sum()))
    elif kind == 4:          # full-scale alternation with a period of 1..40 samples
        p = int(rng.integers(1, 41))
        x = np.where((np.arange(n) // p) % 2 == 0, fs - 1, -fs)
    elif kind == 5:          # full-scale noise
        x = rng.integers(-int(fs), int(fs), size=n).astype(np.float64)
    elif kind == 6:          # random walk
        step = float(rng.uniform(0.5, max(0.6, fs / 64)))
        x = np.cumsum(rng.normal(0, step, size=n))
        x -= x.mean()
    elif kind == 7:          # a dozen close tones
        w0 = float(rng.uniform(0.01, 3.0))
        x = sum(np.cos((w0 + 1e-3 * k * rng.uniform(0.1, 3)) * t + rng.uniform(0, 6.28)) for k in range(12)) * fs / 13.0
    elif kind == 8:          # Nyquist and its neighbours
        x = fs * 0.8 * np.cos(np.pi * t * float(rng.choice([1.0, 0.999, 0.5, 2.0 / 3.0])))
    elif kind == 9:          # wasted bits: multiples of 2^k, k up to bps - 1
        k = int(rng.integers(1, bps))
        x = np.rint(rng.normal(0, fs / 4, size=n) / (1 << k)) * (1 << k)
    elif kind == 10:         # every sample odd
        x = np.rint(rng.normal(0, fs / 8, size=n) / 2) * 2 + 1
    elif kind == 11:         # tiny: -1, 0, 1
        x = rng.integers(-1, 2, size=n).astype(np.float64)
    elif kind == 12:         # a tone plus one LSB of noise (prediction nearly exact)
        x = fs * 0.7 * np.sin(float(rng.uniform(0.005, 1.0)) * t) + rng.integers(-1, 2, size=n)
    elif kind == 13:         # bursts: silence / noise alternating every few hundred samples
        seg = int(rng.integers(50, 700))
        x = rng.normal(0, fs / 6, size=n) * ((np.arange(n) // seg) % 2)
    elif kind == 14:         # a step in the middle of nowhere
        x = np.full(n, float(rng.integers(-int(fs), int(fs))))
        x[int(rng.integers(0, n)):] = float(rng.integers(-int(fs), int(fs)))
    else:                    # decaying noise envelope over the whole signal
        x = rng.normal(0, 1, size=n) * fs * 0.5 * np.exp(-t / max(n / 8.0, 1.0))
    return _fit(x, bps)


def adversarial_signal(rng, n, ch, bps):
    cols = [adversarial_channel(rng, n, bps)]
    for c in range(1, ch):
        rel = int(rng.integers(0, 7))
        a = cols[int(rng.integers(0, c))]
        if rel == 0:
            b = a.copy()
        elif rel == 1:
            b = -a
        elif rel == 2:
            b = np.roll(a, int(rng.integers(1, 40)))
        elif rel == 3:
            b = np.full(n, int(rng.integers(-(1 << (bps - 1)), 1 << (bps - 1))), dtype=np.int64)
        elif rel == 4:
            b = a + rng.integers(-2, 3, size=n)
        else:
            b = adversarial_channel(rng, n, bps)
        cols.append(_fit(b, bps))
    return np.stack(cols, axis=1).astype(np.int32)


def adversarial_case(seed, any_channel_count=False):
    """any_channel_count: the channel count drawn from 1..8 instead of the sweep's 1, 2, 3, 6, 8 (a separate seed space: the cases of
    the soaks on record stay what they were)"""
    import flac_amd
    from test_gpu_parity import _random_config
    rng = np.random.default_rng((990000 if any_channel_count else 770000) + seed)
    while True:
        _, n, ch, bps, rate, kw = _random_config(rng)
        if any_channel_count:
            ch = int(rng.integers(1, 9))
            kw.pop("mid_side", None)
            kw.pop("loose_mid_side", None)
            if ch == 2:
                kw["mid_side"] = int(rng.integers(0, 2))
                kw["loose_mid_side"] = int(rng.integers(0, 2)) if kw["mid_side"] else 0
        n = min(n, 3 * 4608 + 100)                  # (keeps the blocks longer than that to little more than one)
        n = max(n, 1)
        pcm = adversarial_signal(rng, n, ch, bps)
        try:
            s = flac_amd.make_settings(ch, bps, rate, 5, **kw)
        except flac_amd.FlacGpuError:
            continue
        return pcm, ch, bps, rate, kw, s


def ref_kwargs(kw):
    rkw = dict(blocksize=kw["blocksize"], max_lpc_order=kw["max_lpc_order"], streamable_subset=0, min_po=kw["min_partition_order"],
               max_po=kw["max_partition_order"], limit_min_bitrate=kw["limit_min_bitrate"], disable=kw["disable"],
               exhaustive=kw.get("exhaustive", 0), prec_search=kw.get("prec_search", 0))
    if "mid_side" in kw:
        rkw["mid_side"], rkw["loose_mid_side"] = kw["mid_side"], kw["loose_mid_side"]
    if "qlp_coeff_precision" in kw:
        rkw["qlp_precision"] = kw["qlp_coeff_precision"]
    if "apodization" in kw:
        rkw["apodization"] = kw["apodization"]
    return rkw


@pytest.mark.parametrize("any_channel_count", [False, True], ids=["the sweep's channel counts", "1..8 channels"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("FLACGPU_ADV_SEEDS", "24"))))
def test_adversarial_signals_oracle_vs_reference(seed, any_channel_count):
    from oracle_from_settings import oracle_encode_settings
    done = 0
    for sub in range(8):
        pcm, ch, bps, rate, kw, s = adversarial_case(seed * 8 + sub, any_channel_count)
        try:
            r = po.ref_encode(pcm, bps, rate, 5, **ref_kwargs(kw))
        except RuntimeError:
            continue                               # the reference itself gives up (every subframe type disabled on a constant signal)
        o = oracle_encode_settings(pcm, s)
        assert o["data"] == r["data"][r["header_bytes"]:], (seed, sub, ch, bps, rate, kw)
        done += 1
    assert done >= 4


def edge_sum_cases():
    """the fixed-predictor error sums of a quarter block at the edge of 32 bits -- a Nyquist alternation of +-A (|d4| = 16 A)
    at 17..24 bits with A around the amplitude where 1024 fourth differences reach 2^32 (0.5 of full scale at 20 bits, a quarter at
    21, ...), mono / stereo with and without mid/side / three and five channels, blocks of 4096 (fixed.c:222-424: the reference sums
    in 64 bits from 28 - ilog2(n) bits up).  Yields (name, pcm, channels, bps, kwargs for make_settings)."""
    n = 4096 * 3
    sign = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int64)
    for bps in (17, 18, 19, 20, 21, 22, 24):
        fs = 1 << (bps - 1)
        wrap = (1 << 32) / (1024.0 * 16.0)                       # the amplitude at which a quarter's sum of |d4| is 2^32
        for mult in (0.97, 1.0, 1.03, 1.06, 2.02, 0.51):
            a = int(min(fs - 1, round(wrap * mult)))
            if a < 1:
                continue
            for ch, ms in ((1, 0), (2, 0), (2, 1), (3, 0), (5, 0)):
                cols = []
                for c in range(ch):
                    x = sign * a if c % 2 == 0 else -sign * (a - c)   # (the side channel of an anti-phase pair has twice the amplitude)
                    cols.append(x)
                pcm = np.stack(cols, axis=1).astype(np.int32)
                kw = dict(blocksize=4096, streamable_subset=0)
                if ch == 2:
                    kw.update(mid_side=ms, loose_mid_side=0)
                yield "bps%d a%d ch%d ms%d" % (bps, a, ch, ms), pcm, ch, bps, kw


def test_fixed_sums_at_the_32_bit_edge_oracle_vs_reference():
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    for name, pcm, ch, bps, kw in edge_sum_cases():
        for level in (2, 5):
            s = flac_amd.make_settings(ch, bps, 48000, level, **kw)
            rkw = dict(blocksize=4096, streamable_subset=0)
            if ch == 2:
                rkw.update(mid_side=kw["mid_side"], loose_mid_side=0)
            r = po.ref_encode(pcm, bps, 48000, level, **rkw)
            o = oracle_encode_settings(pcm, s)
            assert o["data"] == r["data"][r["header_bytes"]:], (name, level)
