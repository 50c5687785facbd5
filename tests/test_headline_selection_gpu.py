"""-m gpu: the kernels the HEADLINE runs, under the reference-generated goldens and at the size that selects them.

VERDICT r04: the bench's -8 step runs autoc3_kernel<16,13,SETS,PLANES> + prep3_kernel + evalg_kernel + pack2_kernel with the fused
output, but autoc3_kernel is chosen only from 2048 wavefronts up (5462 frames at -8), so the golden digests of the real reference,
the seeded sweep and the full-size configuration tests all ran autoc2_kernel in the driver's suite.  Here:
  * every golden case and 50 seeds of the configuration sweep again with FLACGPU_AUTOC2=1 FLACGPU_AUTOC3=1 (the switches are read
    per context since round 5, flacgpu_api.cpp: read_tune -- no fresh interpreter needed), counting the engines that really
    launched autoc3_kernel (flacgpu_last_batch_kernels).  Round 4's forced test set FLACGPU_AUTOC3=1 alone: at its batch sizes
    launch_analyze never reached launch_autoc2, and autoc3_kernel never ran in it -- found with the kernel record, fixed there too;
  * one 5504-frame -8 batch of the bench signal under the DEFAULT selection, every byte compared with the oracle (not sampled),
    with the kernels the batch launched asserted (flacgpu_last_batch_kernels)."""
import hashlib
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import signals
from cases import golden_cases, case_key, case_pcm, case_search
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as f:
    GOLDEN = json.load(f)
SEEN = {"autoc3": 0, "engines": 0}


def _force_autoc3(monkeypatch, sets=None):
    """the streaming autocorrelation kernels whatever the batch size (FLACGPU_AUTOC2=1: launch_analyze otherwise gives batches of fewer
    than 640 wavefronts to the wavefront-per-job kernel), and of the two the lane-per-subframe one wherever it applies (FLACGPU_AUTOC3=1)"""
    monkeypatch.setenv("FLACGPU_AUTOC2", "1")
    monkeypatch.setenv("FLACGPU_AUTOC3", "1")
    if sets is not None:
        monkeypatch.setenv("FLACGPU_AUTOC3_SETS", str(sets))     # 2: a wavefront per window-job SET whatever the batch size (round 6: small batches go by jobs)


def _encode(pcm, bps, rate, level, max_batch=2048, **kw):
    import flac_amd
    eng = flac_amd.FrameEngine(flac_amd.make_settings(pcm.shape[1], bps, rate, level, **kw), device=0, max_batch_frames=max_batch)
    try:
        data, fb = eng.encode(pcm)
        return data, fb, eng.last_batch_kernels()
    finally:
        eng.close()


@pytest.mark.parametrize("case", golden_cases(), ids=case_key)
def test_reference_golden_with_the_lane_per_subframe_autocorrelation(case, monkeypatch):
    _force_autoc3(monkeypatch, sets=2 if case["level"] % 2 == 0 else 0)       # (both grids: by sets at the even presets, by jobs at the odd ones)
    want = GOLDEN[case_key(case)]
    data, fb, kernels = _encode(case_pcm(case), case["bps"], case["rate"], case["level"], **case_search(case))
    SEEN["engines"] += 1
    SEEN["autoc3"] += "autoc3_kernel" in kernels
    assert len(fb) == want["frames"] and len(data) == want["bytes"]
    assert hashlib.sha256(data).hexdigest() == want["sha256"], sorted(kernels)


def test_the_forced_kernel_really_ran():
    """(after the cases above) stereo streams with a full mid/side search and an LPC search are a good part of the golden set"""
    assert SEEN["engines"] >= 100 and SEEN["autoc3"] >= 20, SEEN


@pytest.mark.parametrize("seed", range(50))
def test_random_configurations_with_the_lane_per_subframe_autocorrelation(seed, monkeypatch):
    import test_gpu_parity as tp
    _force_autoc3(monkeypatch)
    tp.test_random_configurations(seed, monkeypatch)


def _oracle_parallel(pcm, level, block, nthreads=16, chunk=64):
    """the oracle on every frame, chunks of frames on host threads (ctypes releases the GIL), frame numbers as in one stream"""
    nfr = pcm.shape[0] // block
    jobs = [(f, min(f + chunk, nfr)) for f in range(0, nfr, chunk)]

    def one(j):
        lo, hi = j
        o = po.oracle_encode(pcm[lo * block:hi * block], 16, 44100, level, first_frame=lo)
        return o["data"], np.asarray(o["frame_bytes"])
    with ThreadPoolExecutor(nthreads) as ex:
        parts = list(ex.map(one, jobs))
    return b"".join(p[0] for p in parts), np.concatenate([p[1] for p in parts])


def test_full_size_level8_batch_under_the_default_selection_equals_the_oracle_end_to_end():
    """5504 frames x 4096 samples of the bench signal at -8 (>= 5462: the size from which the engine picks autoc3_kernel by itself),
    one engine call, default environment: the kernels launched are the headline's, and all 5504 frames -- 50 MB -- equal the oracle's"""
    import bench
    import flac_amd
    nfr, block = 5504, 4096
    pcm = bench.synth_pcm(nfr, 1234, block)
    for k in ("FLACGPU_AUTOC3", "FLACGPU_AUTOC3_SETS", "FLACGPU_AUTOC3_PLANES", "FLACGPU_NO_FUSED_COMPACT", "FLACGPU_NO_PREP3", "FLACGPU_NO_EVALG"):
        assert k not in os.environ, k
    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, 8), device=0, max_batch_frames=nfr)
    try:
        data, fb = eng.encode(pcm)
        kernels = eng.last_batch_kernels()
        data2, fb2 = eng.encode(pcm)                  # (and again: the second batch finds the first one's tagged words)
        fell = eng.fused_fallbacks()
    finally:
        eng.close()
    # (round 6: a batch of this size -- 344 groups x 3 sets = 1032 wavefronts, half a round of the chip's 2048 slots -- goes a wavefront per
    #  JOB; by sets from 1.5 rounds up: tests/test_large_batch_gpu.py asserts that one)
    want_kernels = {"prep3_kernel", "autoc3_kernel", "autoc3_kernel<PLANES>", "model_kernel", "evalg_kernel", "pack_plan_kernel", "pack2_kernel", "fused_output"}
    assert want_kernels <= kernels and "autoc3_kernel<SETS>" not in kernels, sorted(kernels)
    assert not ({"autoc2_kernel", "autoc_kernel", "prep2_kernel", "prep_kernel", "pack_kernel", "scan_kernel", "compact_kernel"} & kernels), sorted(kernels)
    odata, ofb = _oracle_parallel(pcm, 8, block)
    assert np.array_equal(fb, ofb) and np.array_equal(fb2, ofb)
    assert data == odata and data2 == odata
    # (frames that give up waiting in the fused output cost time, never bytes, and how many do depends on the order in which the chip
    #  starts workgroups -- observed, not promised: none in every run so far, but not a parity condition)
    assert fell[1] <= nfr, fell


@pytest.mark.parametrize("level,nfr,block,want,never", [
    (5, 4608, 4096, {"prep3_kernel", "autoc2_kernel", "evalg_kernel", "pack2_kernel", "fused_output"}, {"autoc3_kernel", "autoc_kernel"}),
    (0, 4608, 1152, {"ff_kernel", "scan_kernel", "compact_kernel"}, {"pack2_kernel", "prep2_kernel"}),
])
def test_other_presets_selection_and_bytes(level, nfr, block, want, never):
    """-5 (one window: autoc2_kernel stays the choice) and -0 (ff_kernel + the two-kernel compaction) at a few thousand frames"""
    import bench
    pcm = bench.synth_pcm(nfr, 77 + level, block)
    data, fb, kernels = _encode(pcm, 16, 44100, level, max_batch=nfr)
    assert want <= kernels and not (never & kernels), sorted(kernels)
    odata, ofb = _oracle_parallel(pcm, level, block)
    assert np.array_equal(fb, ofb) and data == odata


LAYOUTS = [
    ("mono", 1, 16, {}, "music"), ("stereo, independent", 2, 16, dict(mid_side=0), "music"), ("stereo, loose mid/side", 2, 16, dict(mid_side=1, loose_mid_side=1), "music"),
    ("3 channels", 3, 16, {}, "mixed"), ("5.1", 6, 16, {}, "music"), ("8 channels, 24-bit", 8, 24, {}, "music"), ("mono, 24-bit with wasted bits", 1, 24, {}, "wasted"),
    ("stereo 20-bit, independent, one channel with wasted bits", 2, 20, dict(mid_side=0), "music"), ("mono, pure tone", 1, 16, {}, "sine"),
]


@pytest.mark.parametrize("name,ch,bps,kw,fam", LAYOUTS, ids=[l[0] for l in LAYOUTS])
def test_lane_per_subframe_autocorrelation_of_independent_channels(name, ch, bps, kw, fam, monkeypatch):
    """autoc3_kernel<IND> and prep4_kernel (round 5): every channel layout that is not stereo with a full mid/side search -- 64 independent subframes per
    wavefront, each read from its own planar copy (16-bit pairs or 32-bit words, per subframe) -- forced at test size, -8 and -5,
    block sizes whose job lengths leave 1..3 chain steps in the last tile, subframe counts that do not fill the last wavefront; same
    bytes as the oracle, and the kernel record says it ran"""
    _force_autoc3(monkeypatch)
    import flac_amd
    for level, bs, nfr in ((8, 4096, 21), (5, 4096, 13), (8, 4608, 5), (8, 1152, 30), (5, 576, 11)):
        pcm = signals.FAMILIES[fam](bs * nfr + 77, ch, bps)
        if "one channel with wasted bits" in name:
            pcm[:, 1] = (pcm[:, 1] >> 5) << 5
        ekw = dict(kw, blocksize=bs)
        okw = dict(blocksize=bs)
        if "mid_side" in kw:
            okw.update(mid_side=kw["mid_side"], loose=kw.get("loose_mid_side", 0))
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, level, **ekw), device=0, max_batch_frames=64)
        try:
            data, fb = eng.encode(pcm)
            kernels = eng.last_batch_kernels()
        finally:
            eng.close()
        o = po.oracle_encode(pcm, bps, 48000, level, **okw)
        assert "autoc3_kernel<IND>" in kernels, (name, level, bs, sorted(kernels))
        if bs == 4096 and "loose" not in name:
            assert "prep4_kernel" in kernels, (name, level, sorted(kernels))     # (the quarter-per-wavefront prep kernel of independent channels)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, level, bs)


@pytest.mark.parametrize("name,ch,bps,kw,fam", [l for l in LAYOUTS if l[0] in ("mono", "stereo, independent", "5.1", "8 channels, 24-bit", "mono, 24-bit with wasted bits")],
                         ids=[l[0] for l in LAYOUTS if l[0] in ("mono", "stereo, independent", "5.1", "8 channels, 24-bit", "mono, 24-bit with wasted bits")])
def test_independent_channels_with_a_wavefront_per_window_job_set(name, ch, bps, kw, fam, monkeypatch):
    """autoc3_kernel<.., SETS, IND>: a wavefront sweeps the block once -- the whole-block job | the halves | the thirds (-8), the whole | the halves
    (-6, -7) -- for 64 independent subframes; the engine picks it when a wavefront per JOB no longer fits one per SIMD and a wavefront
    per SET still does (mono: 10923..21845 frames).  Forced here at test size, group counts that are not multiples of the XCD count."""
    _force_autoc3(monkeypatch)
    monkeypatch.setenv("FLACGPU_AUTOC3_IND_SETS", "1")
    import flac_amd
    for level, bs, nfr in ((8, 4096, 21), (6, 4096, 150), (7, 4608, 70), (8, 1152, 700), (8, 4096, 577)):
        nfr = max(3, nfr // ch)
        pcm = signals.FAMILIES[fam](bs * nfr + 77, ch, bps)
        ekw = dict(kw, blocksize=bs)
        okw = dict(blocksize=bs)
        if "mid_side" in kw:
            okw.update(mid_side=kw["mid_side"], loose=kw.get("loose_mid_side", 0))
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, level, **ekw), device=0, max_batch_frames=nfr + 1)
        try:
            data, fb = eng.encode(pcm)
            kernels = eng.last_batch_kernels()
        finally:
            eng.close()
        assert {"autoc3_kernel<IND>", "autoc3_kernel<SETS>"} <= kernels, (name, level, bs, sorted(kernels))
        o = po.oracle_encode(pcm, bps, 48000, level, **okw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, level, bs)


@pytest.mark.parametrize("ch,bps", [(1, 16), (2, 16), (6, 16), (8, 24), (3, 12), (1, 24)])
def test_independent_channel_kernels_edge_signals(ch, bps, monkeypatch):
    """prep4_kernel / autoc3_kernel<IND> on the signals that exercise the prep decisions: constant and silent channels (CONSTANT, the
    all-equal test across the four quarters of a frame), limit_min_bitrate (the last channel of an all-constant frame must not be
    CONSTANT), wasted bits per channel, disabled subframe types, a short last block behind the frames of nominal length"""
    _force_autoc3(monkeypatch)
    import flac_amd
    n = 4096 * 9 + 333
    rng = np.random.default_rng(ch * 100 + bps)
    base = signals.music(n, ch, bps, seed=ch + bps)
    variants = []
    v = base.copy(); v[:, -1] = 77; variants.append(("last channel constant", v, {}))
    v = np.zeros_like(base); v[:] = -3; variants.append(("every channel constant", v, dict(limit_min_bitrate=1)))
    v = base.copy(); v[:, 0] = 5; v[4096 * 3 + 1000:, 0] = 6; variants.append(("a step inside a quarter", v, {}))
    v = base.copy(); v[:, 0] = 5; v[4096 * 2 + 2048:, 0] = 9; variants.append(("a step on a quarter's boundary", v, dict(limit_min_bitrate=1)))
    v = (base >> 3) << 3; variants.append(("wasted bits everywhere", v, {}))
    v = base.copy(); v[:, 0] = (v[:, 0] >> 6) << 6; variants.append(("wasted bits in one channel", v, dict(disable=(1, 0, 0))))
    variants.append(("no fixed subframes", base, dict(disable=(0, 1, 0))))
    variants.append(("silence", np.zeros_like(base), dict(disable=(0, 0, 1))))
    from oracle_from_settings import oracle_encode_settings
    for what, pcm, kw in variants:
        s = flac_amd.make_settings(ch, bps, 44100, 8, mid_side=0, **kw)
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=16)
        try:
            data, fb = eng.encode(pcm)
            kernels = eng.last_batch_kernels()
        finally:
            eng.close()
        assert "prep4_kernel" in kernels, (what, sorted(kernels))
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (what, ch, bps)


@pytest.mark.parametrize("order", [16, 17, 20, 24, 25, 31, 32])
def test_plain_autocorrelation_loop_with_a_lane_per_subframe(order, monkeypatch):
    """autoc4_kernel (round 5): from -l 16 up the reference runs the plain loop of lpc.c:133-157 (lag > 16), per lag a sequence of
    additions in increasing sample order -- here with a lane per subframe, every layout from the candidate channels' planes; forced at
    test size (FLACGPU_AUTOC2=1), stereo with mid/side / mono / 5.1 / 24-bit / loose and no mid/side / other block sizes / a pure
    tone / wasted bits; same bytes as the oracle"""
    monkeypatch.setenv("FLACGPU_AUTOC2", "1")
    import flac_amd
    cases = [(2, 40, 16, 4096, "music", {}), (1, 70, 16, 4096, "music", {}), (6, 12, 16, 4096, "music", {}), (2, 21, 24, 4096, "music", {}), (2, 20, 16, 4608, "music", {}),
             (2, 25, 16, 1152, "music", {}), (2, 20, 16, 4096, "sine", {}), (2, 20, 16, 4096, "wasted", {}), (2, 30, 16, 4096, "mixed", dict(mid_side=0)),
             (2, 30, 16, 4096, "music", dict(mid_side=1, loose_mid_side=1)), (2, 33, 16, 576, "music", {})]
    for ch, nfr, bps, bs, fam, kw in cases[order % 3::3] + cases[:1]:
        pcm = signals.FAMILIES[fam](bs * nfr + 55, ch, bps)
        eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, 48000, 8, max_lpc_order=order, streamable_subset=0, blocksize=bs, **kw), device=0, max_batch_frames=nfr + 1)
        try:
            data, fb = eng.encode(pcm)
            kernels = eng.last_batch_kernels()
        finally:
            eng.close()
        okw = dict(max_lpc_order=order, blocksize=bs)
        if "mid_side" in kw:
            okw.update(mid_side=kw["mid_side"], loose=kw.get("loose_mid_side", 0))
        o = po.oracle_encode(pcm, bps, 48000, 8, **okw)
        assert "autoc4_kernel" in kernels and "autoc2_kernel" not in kernels and "autoc3_kernel" not in kernels, sorted(kernels)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (order, ch, bps, bs, fam, kw)


def _oracle_parallel_kw(pcm, bps, rate, level, block, nthreads=16, chunk=32, **okw):
    nfr = pcm.shape[0] // block
    jobs = [(f, min(f + chunk, nfr)) for f in range(0, nfr, chunk)]

    def one(j):
        lo, hi = j
        o = po.oracle_encode(pcm[lo * block:hi * block], bps, rate, level, first_frame=lo, **okw)
        return o["data"], np.asarray(o["frame_bytes"])
    with ThreadPoolExecutor(nthreads) as ex:
        parts = list(ex.map(one, jobs))
    return b"".join(p[0] for p in parts), np.concatenate([p[1] for p in parts])


OFF_THE_FAST_PATH = [
    # name, channels, bps, rate, level, block, frames, engine kw, oracle kw, kernels that must have run
    # (round 6: 2304- and 1152-sample blocks take evalg_kernel and pack2_kernel's 18-sample-run instance; 3456 = 64 runs of 54 still has the
    #  general evaluation body -- and the 18-sample-run pack instance, three passes of it; the general pack body: the -l 32 case below)
    ("-8 -b 3456: general evaluation body, pack2<run18>", 2, 16, 44100, 8, 3456, 2000, dict(blocksize=3456, streamable_subset=0), dict(blocksize=3456), {"eval_kernel", "pack2_kernel<run18>"}),
    ("-8 -b 2304: evalg + pack2<run18>", 2, 16, 44100, 8, 2304, 3000, dict(blocksize=2304), dict(blocksize=2304), {"evalg_kernel", "pack2_kernel<run18>", "fused_output"}),
    ("-8 -b 1152: evalg + pack2<run18>", 2, 16, 44100, 8, 1152, 6000, dict(blocksize=1152), dict(blocksize=1152), {"evalg_kernel", "pack2_kernel<run18>", "fused_output"}),
    ("-8 -l 32: autoc4 + the general evaluation, model and pack", 2, 16, 44100, 8, 4096, 1200, dict(max_lpc_order=32, streamable_subset=0), dict(max_lpc_order=32), {"autoc4_kernel", "eval_kernel", "pack_kernel", "scan_kernel", "compact_kernel"}),
    # (round 6: 13..32 taps on evalg_kernel<32>; 17..24-bit stereo at -0..-2 in ff_kernel<., ., WIDE>)
    ("-8 -l 16: autoc4 + evalg_kernel<32> + pack2", 2, 16, 44100, 8, 4096, 2500, dict(max_lpc_order=16, streamable_subset=0), dict(max_lpc_order=16), {"autoc4_kernel", "evalg_kernel", "pack2_kernel"}),
    ("-2 on 24-bit: the one-kernel frame (wide)", 2, 24, 96000, 2, 1152, 6000, {}, {}, {"ff_kernel", "scan_kernel", "compact_kernel"}),
    ("-0 on 24-bit mono", 1, 24, 48000, 0, 1152, 6000, {}, {}, {"prep2_kernel<DECIDE>", "pack2_kernel"}),
    ("-8 mono under the default selection", 1, 16, 44100, 8, 4096, 9000, {}, {}, {"prep4_kernel", "autoc3_kernel<IND>", "evalg_kernel", "pack2_kernel"}),
    ("-8 mono, 11264 frames: the size at which independent channels go a wavefront per window-job set", 1, 16, 44100, 8, 4096, 11264, {}, {}, {"prep4_kernel", "autoc3_kernel<IND>", "autoc3_kernel<SETS>", "evalg_kernel", "pack2_kernel"}),
    ("-8 5.1 under the default selection", 6, 16, 48000, 8, 4096, 1800, {}, {}, {"prep4_kernel", "autoc3_kernel<IND>", "evalg_kernel", "pack2_kernel"}),
    ("-5 stereo without mid/side", 2, 16, 44100, 5, 4096, 5000, dict(mid_side=0), dict(mid_side=0), {"prep4_kernel", "evalg_kernel", "pack2_kernel"}),
]


@pytest.mark.parametrize("name,ch,bps,rate,level,block,nfr,ekw,okw,want", OFF_THE_FAST_PATH, ids=[c[0] for c in OFF_THE_FAST_PATH])
def test_batches_of_thousands_of_frames_off_the_headline_path(name, ch, bps, rate, level, block, nfr, ekw, okw, want):
    """VERDICT r04, thin spot (iii): what is not the headline's path -- the general evaluation / model / pack bodies, the kernels of the
    other channel layouts, the plain autocorrelation loop -- saw test-sized batches only.  Here each runs a batch of thousands of
    frames under the engine's own selection (no switches), every byte compared with the oracle (all frames, host threads), the
    kernels that ran asserted."""
    import flac_amd
    base = signals.music(block * 96, ch, bps, seed=len(name)).astype(np.int64)
    reps = -(-nfr // 96)
    pcm = np.concatenate([(base * (1000 - 3 * (r % 150))) // 1000 + (r % 3) for r in range(reps)])[:nfr * block].astype(np.int32)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, bps, rate, level, **ekw), device=0, max_batch_frames=nfr)
    try:
        data, fb = eng.encode(pcm)
        kernels = eng.last_batch_kernels()
    finally:
        eng.close()
    assert want <= kernels, (name, sorted(kernels))
    odata, ofb = _oracle_parallel_kw(pcm, bps, rate, level, block, **okw)
    assert np.array_equal(fb, ofb), name
    assert data == odata, name


@pytest.mark.parametrize("ch,bps,level,kw", [(1, 16, 8, {}), (3, 16, 5, {}), (2, 24, 8, dict(mid_side=0)), (1, 16, 8, dict(max_lpc_order=16, streamable_subset=0))])
def test_the_last_group_of_independent_subframes_is_whole(ch, bps, level, kw, monkeypatch):
    """round 6: autoc3_kernel<IND> / autoc4_kernel start the batch's last group of 64 rows 64 rows before the end (a partial group took
    the slow fetch in every window job).  Subframe counts around the multiples of 64 -- fewer than a group, a whole number of groups, one
    row more, one row less -- against the oracle with the kernel asserted; rows shared with the group before are computed twice."""
    _force_autoc3(monkeypatch)
    monkeypatch.setenv("FLACGPU_POISON", "1")
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    N = 1024
    want = "autoc4_kernel" if kw.get("max_lpc_order") == 16 else "autoc3_kernel<IND>"
    st = flac_amd.make_settings(ch, bps, 48000, level, blocksize=N, **kw)
    for rows in (1, 63, 64, 65, 127, 128, 129, 190):
        nframes = (rows + ch - 1) // ch
        pcm = signals.music(nframes * N, ch, bps, seed=rows)
        eng = flac_amd.FrameEngine(st, device=0, max_batch_frames=nframes)
        try:
            data, fb = eng.encode(pcm)
            kernels = eng.last_batch_kernels()
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, st)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (rows, ch, bps, level, kw, sorted(kernels))
        if nframes * ch >= 2:
            assert want in kernels, (rows, sorted(kernels))
