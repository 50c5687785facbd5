"""Case list shared by make_golden.py (generator, needs oracle/_ref) and the tests."""

def golden_cases():
    cases = []
    n16 = 4096 * 3 + 1000
    for fam in ["music", "white", "sine", "constant", "silence", "wasted", "square", "quiet", "mixed"]:
        for level in range(9):
            cases.append(dict(family=fam, n=n16, channels=2, bps=16, rate=44100, level=level))
    for level in (0, 3, 5, 8):
        cases.append(dict(family="music", n=4096 * 2 + 123, channels=2, bps=24, rate=96000, level=level))
        cases.append(dict(family="music", n=4096 * 2 + 77, channels=1, bps=16, rate=44100, level=level))
        cases.append(dict(family="white", n=4096 * 2 + 5, channels=2, bps=24, rate=96000, level=level))
    for tail in (1, 4, 5, 17, 32, 33, 100, 1365, 2049, 4095):
        for level in (2, 5, 8):
            cases.append(dict(family="music", n=(1152 if level < 3 else 4096) + tail, channels=2, bps=16, rate=44100, level=level))
    for ch in (3, 6):
        cases.append(dict(family="music", n=4096 + 50, channels=ch, bps=16, rate=48000, level=8))
    for bps in (8, 12, 20):
        cases.append(dict(family="music", n=4096 * 2 + 9, channels=2, bps=bps, rate=32000, level=6))
    # pure tones: the ill-conditioned case that pins the autocorrelation association order
    for freq in (441.0, 1000.0, 997.0, 11025.0):
        for level in (5, 8):
            cases.append(dict(family="sine", n=4096 * 4, channels=1, bps=16, rate=44100, level=level, freq=freq))
    # the wider model searches: -e (every fixed / LPC order) and -p (every coefficient precision)
    for level in (3, 5, 8):
        for ex, ps in ((1, 0), (0, 1), (1, 1)):
            cases.append(dict(family="music", n=4096 * 2 + 411, channels=2, bps=16, rate=44100, level=level, exhaustive=ex, prec_search=ps))
    for ex, ps in ((1, 0), (1, 1)):
        cases.append(dict(family="mixed", n=4096 * 2 + 33, channels=2, bps=16, rate=44100, level=8, exhaustive=ex, prec_search=ps))
        cases.append(dict(family="music", n=4096 + 200, channels=2, bps=24, rate=96000, level=8, exhaustive=ex, prec_search=ps))
        cases.append(dict(family="sine", n=4096 * 2, channels=1, bps=16, rate=44100, level=5, exhaustive=ex, prec_search=ps))
    # the widened range: more than 24 bits per sample (33-bit side channel at 32), orders above 15, long blocks, frames larger
    # than the LDS, full-scale deflection, slow anti-phase pairs (overflow-checked residuals at 24 bits)
    for bps in (28, 32):
        for fam in ("music", "white"):
            for level in (5, 8):
                cases.append(dict(family=fam, n=4096 * 2 + 300, channels=2, bps=bps, rate=96000, level=level, lax=1))
    for pattern in (1, 3, 6):
        cases.append(dict(family="fsd", n=4096 + 500, channels=2, bps=32, rate=96000, level=5, pattern=pattern))
        cases.append(dict(family="fsd", n=4096 + 500, channels=1, bps=16, rate=44100, level=8, pattern=pattern))
    cases.append(dict(family="music", n=4096 * 2 + 99, channels=2, bps=16, rate=96000, level=8, max_lpc_order=32))
    cases.append(dict(family="music", n=4096 + 99, channels=2, bps=24, rate=96000, level=5, max_lpc_order=20, exhaustive=1))
    cases.append(dict(family="sine", n=4096 * 2, channels=1, bps=16, rate=96000, level=8, max_lpc_order=16))
    cases.append(dict(family="music", n=65535 + 4000, channels=2, bps=16, rate=44100, level=5, blocksize=65535, lax=1))
    cases.append(dict(family="mixed", n=32768 * 2 + 11, channels=1, bps=24, rate=96000, level=8, blocksize=32768, lax=1))
    cases.append(dict(family="music", n=16384 * 2 + 700, channels=6, bps=16, rate=96000, level=5, blocksize=16384))
    for kind in (0, 3):
        cases.append(dict(family="slow", n=4096 * 2 + 77, channels=2, bps=24, rate=96000, level=8, kind=kind))
    return cases


def case_search(c, for_oracle=False):
    """keyword arguments beyond the preset, for pyoracle / the reference shim / make_settings alike (the oracle has no
    notion of the streamable subset: for_oracle drops that switch)"""
    kw = dict(exhaustive=c.get("exhaustive", 0), prec_search=c.get("prec_search", 0))
    for k in ("max_lpc_order", "blocksize"):
        if k in c:
            kw[k] = c[k]
    if c.get("lax") and not for_oracle:
        kw["streamable_subset"] = 0
    return kw


def case_key(c):
    return "|".join("%s=%s" % (k, c[k]) for k in sorted(c))


def case_pcm(c):
    import signals
    kw = {}
    if "freq" in c:
        kw["freq"] = c["freq"]
    if c["family"] == "fsd":
        return signals.fsd(c["n"], c["channels"], c["bps"], c["pattern"])
    if c["family"] == "slow":
        return signals.slow(c["n"], c["channels"], c["bps"], c["kind"])
    return signals.FAMILIES[c["family"]](c["n"], c["channels"], c["bps"], **kw)
