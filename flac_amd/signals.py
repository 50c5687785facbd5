"""Deterministic synthetic PCM generators shared by the tests and bench.py.

Every generator returns int32 [nsamples, channels] within the given bit depth.
The families follow SURVEY.md section 8d: white noise (config 2), "music-like" tones +
coloured noise with inter-channel correlation (config 3/4), pure sines (config 1, the
ill-conditioned case for the autocorrelation reduction order), plus the edge cases the
reference's test_streams generator covers (constant, silence, wasted bits, full-scale).
"""
import numpy as np


def _clip(x, bps):
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    return np.clip(np.rint(x), lo, hi).astype(np.int32)


def white(n, channels=2, bps=16, seed=1234):
    rng = np.random.default_rng(seed)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1))
    return rng.integers(lo, hi, size=(n, channels), dtype=np.int64).astype(np.int32)


def music(n, channels=2, bps=16, seed=1234, rate=44100):
    """2-3 sines + FIR-coloured Gaussian noise, scaled to ~0.6 full scale, channels correlated."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / rate
    amp = 0.6 * (1 << (bps - 1))
    base = 0.45 * np.sin(2 * np.pi * 441.0 * t) + 0.25 * np.sin(2 * np.pi * 1234.5 * t + 0.3)
    base += 0.1 * np.sin(2 * np.pi * 2000.0 * t + 1.1) * (1.0 + 0.5 * np.sin(2 * np.pi * 0.7 * t))
    k = np.array([1, 2, 3, 4, 3, 2, 1], dtype=np.float64)
    k /= k.sum()
    out = np.empty((n, channels), dtype=np.int32)
    common = np.convolve(rng.standard_normal(n + 6), k, mode="valid")
    for c in range(channels):
        own = np.convolve(rng.standard_normal(n + 6), k, mode="valid")
        x = base * (1.0 - 0.15 * c) + 0.12 * common + 0.05 * own
        out[:, c] = _clip(amp * x, bps)
    return out


def sine(n, channels=1, bps=16, freq=441.0, rate=44100, amp=0.9, phase=0.0):
    t = np.arange(n, dtype=np.float64)
    x = amp * ((1 << (bps - 1)) - 1) * np.sin(2 * np.pi * freq * t / rate + phase)
    out = np.empty((n, channels), dtype=np.int32)
    for c in range(channels):
        out[:, c] = _clip(x if c == 0 else 0.5 * x, bps)
    return out


def constant(n, channels=2, bps=16, value=1234):
    return np.full((n, channels), value, dtype=np.int32)


def silence(n, channels=2, bps=16):
    return np.zeros((n, channels), dtype=np.int32)


def wasted(n, channels=2, bps=16, seed=7, shift=3):
    """low `shift` bits always zero -> wasted-bits path (stream_encoder.c:5077)"""
    x = music(n, channels, bps - shift, seed)
    return (x << shift).astype(np.int32)


def fullscale_square(n, channels=2, bps=16, period=37):
    hi, lo = (1 << (bps - 1)) - 1, -(1 << (bps - 1))
    t = (np.arange(n) // period) % 2
    x = np.where(t == 0, hi, lo).astype(np.int32)
    return np.stack([x if c % 2 == 0 else -1 - x for c in range(channels)], axis=1).astype(np.int32)


def quiet(n, channels=2, bps=16, seed=3):
    """very low level noise: Rice parameter 0/1 region"""
    rng = np.random.default_rng(seed)
    return rng.integers(-2, 3, size=(n, channels)).astype(np.int32)


def mixed(n, channels=2, bps=16, seed=99):
    """concatenation of regimes so one stream touches constant/verbatim/fixed/LPC frames"""
    parts = [music(n // 4, channels, bps, seed), silence(n // 8, channels, bps),
             white(n // 8, channels, bps, seed + 1), sine(n // 4, channels, bps),
             wasted(n // 8, channels, bps, seed + 2), quiet(n // 8, channels, bps, seed + 3)]
    x = np.concatenate(parts, axis=0)
    if x.shape[0] < n:
        x = np.concatenate([x, constant(n - x.shape[0], channels, bps, 77)], axis=0)
    return x[:n]


def slow(n, channels=2, bps=24, kind=0, rate=96000):
    """very smooth full-scale signals, the two channels in anti-phase: the LPC coefficients grow past 2^7, the predicted
    residual width past 32 bits, and the reference switches to its overflow-checked FIR (stream_encoder.c:4601-4609)"""
    t = np.arange(n, dtype=np.float64)
    if kind == 0:
        x = np.sin(2 * np.pi * 2.0 * t / rate)
    elif kind == 1:
        x = sum(np.sin(2 * np.pi * f * t / rate + f) for f in (3.0, 7.0, 11.0, 19.0, 31.0, 43.0)) / 6
    elif kind == 2:
        x = (t / n - 0.5) ** 7
    else:
        x = np.sin(2 * np.pi * (1.0 + 30.0 * t / n) * t / rate)
    x = x / np.abs(x).max() * ((1 << (bps - 1)) - 1) * 0.99
    cols = [np.round(x) if c % 2 == 0 else np.round(-x * 0.98) for c in range(channels)]
    return np.stack(cols, axis=1).astype(np.int32)


FSD_PATTERNS = {1: (1, -1), 2: (1, 1, -1), 3: (1, -1, -1), 4: (1, -1, 1, -1), 5: (1, -1, -1, 1), 6: (1, -1, 1, 1, -1), 7: (1, -1, -1, 1, -1)}


def fsd(n, channels=1, bps=16, pattern=1):
    """full-scale deflection streams of the reference's test suite (src/test_streams/main.c:306-433,1341-1347);
    the second channel of a stereo pair runs the pattern inverted, which drives the side channel to bps+1 bits"""
    hi, lo = (1 << (bps - 1)) - 1, -(1 << (bps - 1))
    pat = np.array([hi if v > 0 else lo for v in FSD_PATTERNS[pattern]], dtype=np.int64)
    x = np.resize(pat, n)
    return np.stack([x if c % 2 == 0 else -1 - x for c in range(channels)], axis=1).astype(np.int32)


FAMILIES = {
    "white": white, "music": music, "sine": sine, "constant": constant, "silence": silence,
    "wasted": wasted, "square": fullscale_square, "quiet": quiet, "mixed": mixed,
}
