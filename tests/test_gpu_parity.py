"""-m gpu: the HIP path (through the C ABI, libflacgpu.so) against the oracle, the committed golden
digests of the real reference, and -- at BASELINE.json's full sizes -- size-independent properties.
Bit-exact is the bar: integer/byte output."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import signals
from cases import golden_cases, case_key, case_pcm, case_search
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as f:
    GOLDEN = json.load(f)


def _engine(channels, bps, rate, level, max_batch=2048, **kw):
    import flac_amd
    return flac_amd.FrameEngine(flac_amd.make_settings(channels, bps, rate, level, **kw), device=0, max_batch_frames=max_batch)


def _gpu_encode(pcm, bps, rate, level, first_frame=0, max_batch=2048, **kw):
    eng = _engine(pcm.shape[1], bps, rate, level, max_batch, **kw)
    try:
        return eng.encode(pcm, first_frame)
    finally:
        eng.close()


def test_extension_is_loaded_and_device_present():
    import flac_amd
    from flac_amd import engine
    lib = engine.load_engine()
    assert lib.flacgpu_device_count() >= 1
    assert os.path.samefile(engine.ENGINE_SO, os.path.join(os.path.dirname(flac_amd.__file__), "lib", "libflacgpu.so"))


@pytest.mark.parametrize("case", golden_cases(), ids=case_key)
def test_gpu_matches_reference_golden(case):
    want = GOLDEN[case_key(case)]
    data, fb = _gpu_encode(case_pcm(case), case["bps"], case["rate"], case["level"], **case_search(case))
    assert len(fb) == want["frames"]
    assert len(data) == want["bytes"]
    assert hashlib.sha256(data).hexdigest() == want["sha256"]


@pytest.mark.parametrize("level", range(9))
def test_gpu_matches_oracle_per_frame(level):
    for fam in ("music", "mixed", "white"):
        pcm = signals.FAMILIES[fam](4096 * 9 + 1234, 2, 16, seed=level + 1)
        data, fb = _gpu_encode(pcm, 16, 44100, level)
        o = po.oracle_encode(pcm, 16, 44100, level)
        assert np.array_equal(fb, o["frame_bytes"])
        assert data == o["data"]


def test_gpu_subframe_decisions_match_oracle():
    """not only the bytes: the per-subframe model choices reported by the engine equal the oracle's"""
    pcm = signals.mixed(4096 * 8, 2, 16)
    eng = _engine(2, 16, 44100, 8)
    data, fb = eng.encode(pcm)
    sub, ca = eng.last_batch_info(8)
    lib = po.load_oracle()
    cfg = po.OracleConfig(2, 16, 44100, 8)
    planar = np.ascontiguousarray(pcm.T)
    out = np.empty(1 << 17, np.uint8)
    for f in range(8):
        info = po.FoFrameInfo()
        ptrs = (C.c_void_p * 2)(planar[0, f * 4096:].ctypes.data, planar[1, f * 4096:].ctypes.data)
        assert lib.fo_encode_frame(C.byref(cfg.c), ptrs, f, out.ctypes.data, out.size, C.byref(info)) == fb[f]
        assert info.channel_assignment == ca[f]
        for ch in range(2):
            g, o = sub[f * 2 + ch], info.sub[ch]
            assert (g.type, g.wasted_bits, g.bits) == (o.type, o.wasted_bits, o.bits)
            if o.type >= 2:
                assert (g.order, g.partition_order, g.rice2) == (o.order, o.partition_order, o.rice2)
            if o.type == 3:
                assert (g.precision, g.shift) == (o.precision, o.shift)
    eng.close()


@pytest.mark.parametrize("tail", [1, 4, 5, 31, 32, 33, 100, 683, 1365, 1932, 2047, 3860, 4095])
def test_short_last_block(tail):
    for level, bps, rate in ((2, 16, 44100), (5, 16, 44100), (8, 16, 44100), (8, 24, 96000)):
        n = (1152 if level < 3 else 4096) * 2 + tail
        pcm = signals.music(n, 2, bps, seed=tail)
        data, fb = _gpu_encode(pcm, bps, rate, level)
        assert data == po.oracle_encode(pcm, bps, rate, level)["data"], (tail, level, bps)


def test_batches_and_frame_numbers():
    """a stream split over several engine calls equals one call; frame numbers cross UTF-8 length classes"""
    pcm = signals.music(4096 * 37 + 99, 2, 16, seed=21)
    one, fb1 = _gpu_encode(pcm, 16, 44100, 5, max_batch=64)
    many, fb2 = _gpu_encode(pcm, 16, 44100, 5, max_batch=5)
    assert one == many and np.array_equal(fb1, fb2)
    # every UTF-8 length class of the frame number, each crossed inside the batch: 1->2, 2->3, 3->4, 4->5, 5->6 bytes
    # (bitwriter.c:810-845; 0x4000000 opens the 6-byte class, 0x7ffffffe is the last pair of 31-bit numbers)
    for first in (0x7E, 0x7FE, 0xFFFE, 0x1FFFFE, 0x3FFFFFE, 0x7FFFFFFC):
        data, _ = _gpu_encode(pcm[:4096 * 4], 16, 44100, 5, first_frame=first)
        assert data == po.oracle_encode(pcm[:4096 * 4], 16, 44100, 5, first_frame=first)["data"]


@pytest.mark.parametrize("level", [0, 1, 2, 8])
def test_frame_numbers_of_the_one_kernel_path_and_the_fused_output(level):
    """ff_kernel (-0 .. -2) and pack2_kernel with the fused output (-8) write their own frame headers: every UTF-8 length class of the
    frame number, each crossed inside the batch (only the -5 path had this test), with a short last block behind"""
    bs = 1152 if level < 3 else 4096
    pcm = signals.music(bs * 5 + 77, 2, 16, seed=31 + level)
    for first in (0x7E, 0x7FE, 0xFFFE, 0x1FFFFE, 0x3FFFFFE, 0x7FFFFFFA):
        data, fb = _gpu_encode(pcm, 16, 44100, level, first_frame=first)
        o = po.oracle_encode(pcm, 16, 44100, level, first_frame=first)
        assert data == o["data"] and np.array_equal(fb, o["frame_bytes"]), (level, hex(first))


def test_limit_min_bitrate_and_channel_counts():
    for level in (0, 2, 5, 8):
        for pcm in (signals.silence(4096 * 3, 2, 16), signals.mixed(4096 * 6, 2, 16), signals.silence(4096 * 2, 1, 16)):
            data, _ = _gpu_encode(pcm, 16, 44100, level, limit_min_bitrate=1)
            assert data == po.oracle_encode(pcm, 16, 44100, level, limit_min_bitrate=1)["data"]
    for ch in (1, 3, 4, 8):
        pcm = signals.music(4096 * 2 + 50, ch, 16, seed=ch)
        data, _ = _gpu_encode(pcm, 16, 48000, 8)
        assert data == po.oracle_encode(pcm, 16, 48000, 8)["data"]


def test_config2_1000_noise_frames_level5():
    """BASELINE.json configs[1]: flac -5 on 1000 x 4096-sample 44.1k/16-bit stereo white-noise frames"""
    pcm = signals.white(1000 * 4096, 2, 16, seed=1234)
    data, fb = _gpu_encode(pcm, 16, 44100, 5, max_batch=1000)
    o = po.oracle_encode(pcm, 16, 44100, 5)
    assert len(fb) == 1000 and np.array_equal(fb, o["frame_bytes"])
    assert hashlib.sha256(data).hexdigest() == hashlib.sha256(o["data"]).hexdigest()


@pytest.mark.parametrize("bps,rate", [(16, 44100), (24, 96000)])
def test_config3_4_level8_full_size_and_roundtrip(bps, rate):
    """configs[2]/[3]: flac -8 at full batch size; compare with the oracle AND decode the GPU's frames with the
    reference decoder (when oracle/_ref is present) -- encode->decode identity is size independent."""
    nframes = 1500
    pcm = signals.music(nframes * 4096, 2, bps, seed=77, rate=rate)
    data, fb = _gpu_encode(pcm, bps, rate, 8, max_batch=nframes)
    o = po.oracle_encode(pcm, bps, rate, 8)
    assert np.array_equal(fb, o["frame_bytes"])
    assert data == o["data"]
    # every frame carries a valid CRC-16 footer (checksum of checksums style property)
    lib = po.load_oracle()
    off = np.concatenate([[0], np.cumsum(fb.astype(np.int64))])
    buf = np.frombuffer(data, dtype=np.uint8)
    for f in range(0, nframes, 97):
        frame = buf[off[f]:off[f + 1]]
        assert lib.fo_crc16(frame.ctypes.data, len(frame) - 2) == int(frame[-2]) << 8 | int(frame[-1])
    if po.have_ref():
        _decode_and_compare(data, pcm, bps, rate)


def _decode_and_compare(frames, pcm, bps, rate):
    """minimal fLaC + STREAMINFO wrapper, then the reference's stream decoder (test infrastructure)."""
    ref = po.load_ref()
    n, ch = pcm.shape
    si = bytearray(34)
    si[0:2] = (4096).to_bytes(2, "big"); si[2:4] = (4096).to_bytes(2, "big")
    v = (rate << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | n
    si[10:18] = v.to_bytes(8, "big")
    blob = b"fLaC" + bytes([0x80, 0, 0, 34]) + bytes(si) + frames
    state = {"pos": 0, "out": [], "err": 0}
    RD = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t), C.c_void_p)
    WR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
    ER = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

    def rd(dec, buf, nbytes, cd):
        want = nbytes[0]
        chunk = blob[state["pos"]:state["pos"] + want]
        if not chunk:
            nbytes[0] = 0
            return 1   # END_OF_STREAM
        C.memmove(buf, chunk, len(chunk))
        nbytes[0] = len(chunk)
        state["pos"] += len(chunk)
        return 0

    def wr(dec, frame, buffers, cd):
        bs = C.cast(frame, C.POINTER(C.c_uint32))[0]     # FLAC__Frame.header.blocksize is the first field
        state["out"].append(np.stack([np.ctypeslib.as_array(buffers[c], (bs,)).copy() for c in range(ch)], axis=1))
        return 0

    def er(dec, status, cd):
        state["err"] += 1

    cbs = (RD(rd), WR(wr), ER(er))
    ref.FLAC__stream_decoder_new.restype = C.c_void_p
    dec = C.c_void_p(ref.FLAC__stream_decoder_new())
    ref.FLAC__stream_decoder_init_stream.argtypes = [C.c_void_p, RD, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, WR, C.c_void_p, ER, C.c_void_p]
    assert ref.FLAC__stream_decoder_init_stream(dec, cbs[0], None, None, None, None, cbs[1], None, cbs[2], None) == 0
    ref.FLAC__stream_decoder_process_until_end_of_stream.argtypes = [C.c_void_p]
    assert ref.FLAC__stream_decoder_process_until_end_of_stream(dec)
    ref.FLAC__stream_decoder_finish.argtypes = [C.c_void_p]
    ref.FLAC__stream_decoder_finish(dec)
    ref.FLAC__stream_decoder_delete.argtypes = [C.c_void_p]
    ref.FLAC__stream_decoder_delete(dec)
    assert state["err"] == 0
    got = np.concatenate(state["out"], axis=0)
    assert got.shape == pcm.shape and np.array_equal(got, pcm)


def test_device_resident_entry_point():
    """flacgpu_encode_batch_device: PCM, frames and lengths never leave HBM; same bytes as the host entry"""
    import torch
    pcm = signals.music(4096 * 64, 2, 16, seed=5)
    eng = _engine(2, 16, 44100, 8, max_batch=64)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = eng.max_output_bytes(64)
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.zeros(64, dtype=torch.int32, device=dev)
    d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
    eng.encode_device(d_pcm.data_ptr(), 64, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
    torch.cuda.synchronize()
    total = int(d_tot.item())
    o = po.oracle_encode(pcm, 16, 44100, 8)
    assert total == len(o["data"])
    assert d_out[:total].cpu().numpy().tobytes() == o["data"]
    assert np.array_equal(d_fb.cpu().numpy().astype(np.uint32), o["frame_bytes"])
    a, p, c = eng.last_kernel_ms()
    assert a > 0 and p > 0
    eng.close()


@pytest.mark.parametrize("level,block", [(8, 4096), (5, 4096), (0, 1152), (2, 1152), (8, 2304)])
@pytest.mark.parametrize("copy", [0, 1])
def test_device_entry_writes_lengths_and_total_into_the_callers_arrays(level, block, copy, monkeypatch):
    """round 5: the kernels write the frame lengths and the stream's length where the caller asked for them (FLACGPU_COPY_RESULTS=1:
    round 4's two copies behind the last kernel) -- arrays in the middle of larger ones (4- and 8-byte aligned, nothing beyond the
    batch's entries touched), two batches back to back into different arrays, a short last block, every preset family's last kernel
    (fused output, scan + compact behind ff_kernel, the general pack kernel)"""
    import torch
    import flac_amd
    if copy:
        monkeypatch.setenv("FLACGPU_COPY_RESULTS", "1")
    nfr, tail = 37, 333
    pcm = signals.music(block * nfr + tail, 2, 16, seed=level + block)
    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, level, blocksize=block), device=0, max_batch_frames=64)
    try:
        dev = torch.device("cuda", 0)
        d_pcm = torch.from_numpy(pcm).to(dev)
        cap = eng.max_output_bytes(nfr + 1)
        outs = []
        for k, first in enumerate((0, 1000)):
            d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
            d_fb = torch.full((nfr + 1 + 6,), -7, dtype=torch.int32, device=dev)
            d_tot = torch.full((4,), -9, dtype=torch.int64, device=dev)
            eng.encode_device(d_pcm.data_ptr(), nfr + 1, d_out.data_ptr(), cap, d_fb.data_ptr() + 4 * (1 + 2 * k), d_tot.data_ptr() + 8 * (1 + k), first_frame_number=first, tail=tail)
            outs.append((d_out, d_fb, d_tot, first, 1 + 2 * k, 1 + k))
        torch.cuda.synchronize()
        for d_out, d_fb, d_tot, first, fo, to in outs:
            o = po.oracle_encode(pcm, 16, 44100, level, blocksize=block, first_frame=first)
            fb, tot = d_fb.cpu().numpy(), d_tot.cpu().numpy()
            assert int(tot[to]) == len(o["data"]) and all(int(tot[i]) == -9 for i in range(4) if i != to), tot
            assert np.array_equal(fb[fo:fo + nfr + 1].astype(np.uint32), o["frame_bytes"])
            assert np.all(fb[:fo] == -7) and np.all(fb[fo + nfr + 1:] == -7), fb
            assert d_out[:len(o["data"])].cpu().numpy().tobytes() == o["data"]
    finally:
        eng.close()


# ---- input staging on the device (format_input of the reference's CLI, src/flac/encode.c:2352-2492) ---------------------
RAW_FORMATS = [(8, False, True, 0), (8, False, False, 0), (16, False, False, 0), (16, True, False, 0), (16, False, True, 0),
               (24, False, False, 0), (24, True, False, 0), (24, True, True, 0), (32, False, False, 8), (32, True, True, 8),
               (16, False, False, 4), (24, True, False, 4)]


@pytest.mark.parametrize("fmt", RAW_FORMATS, ids=lambda f: "c%d_%s_%s_sh%d" % (f[0], "be" if f[1] else "le", "u" if f[2] else "s", f[3]))
def test_raw_staging_matches_format_input_and_encodes_identically(fmt):
    import torch
    import flac_amd
    from rawfmt import format_input, to_raw
    bits, be, uns, shift = fmt
    bps = bits - shift
    channels = 2
    pcm = signals.music(4096 * 3 + 777, channels, bps, seed=bits + shift)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    pcm = np.clip(pcm, lo, hi).astype(np.int32)
    cmap = [1, 0] if bits == 24 else None
    raw = to_raw(pcm, bits, be, uns, shift, cmap)
    assert np.array_equal(format_input(raw, channels, bits, be, uns, shift, cmap), pcm)
    eng = _engine(channels, bps, 44100, 5, max_batch=16)
    try:
        f = flac_amd.raw_format(bits, be, uns, shift, cmap)
        # the staging kernel alone, device to device
        d_raw = torch.from_numpy(raw.copy()).cuda()
        d_pcm = torch.empty(pcm.size, dtype=torch.int32, device="cuda")
        d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
        eng.stage_raw_device(d_raw.data_ptr(), f, pcm.shape[0], d_pcm.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        assert int(d_err.item()) == 0
        assert np.array_equal(d_pcm.cpu().numpy().reshape(pcm.shape), pcm)
        # raw host bytes -> frames == int32 host block -> frames
        want, wfb = eng.encode(pcm)
        got, gfb = eng.encode_raw(raw, f)
        assert np.array_equal(gfb, wfb) and got == want
    finally:
        eng.close()


def test_raw_staging_reports_shift_violation():
    import flac_amd
    from rawfmt import to_raw
    pcm = signals.music(4096, 2, 12, seed=5).astype(np.int32)
    raw = to_raw(np.clip(pcm, -2048, 2047), 16, False, False, 4).copy()
    raw[10] |= 1                                       # a bit below the declared shift
    eng = _engine(2, 12, 44100, 5, max_batch=4)
    try:
        with pytest.raises(flac_amd.FlacGpuError, match="non-zero bits"):
            eng.encode_raw(raw, flac_amd.raw_format(16, False, False, 4))
    finally:
        eng.close()


@pytest.mark.parametrize("mode", [(1, 0), (0, 1), (1, 1)], ids=["e", "p", "ep"])
@pytest.mark.parametrize("level", [0, 2, 3, 4, 6, 7, 8])
def test_wider_model_searches_match_oracle(level, mode):
    """-e / -p (stream_encoder.c:4155-4163, 4220-4243): every frame equal to the oracle's, incl. a short last block"""
    ex, ps = mode
    for fam, bps, rate in (("music", 16, 44100), ("mixed", 16, 44100), ("wasted", 16, 44100), ("music", 24, 96000)):
        if bps == 24 and level not in (3, 8):
            continue
        pcm = signals.FAMILIES[fam](4096 * 2 + 700, 2, bps, seed=level + 3) if fam != "wasted" else signals.FAMILIES[fam](4096 * 2 + 700, 2, bps)
        data, fb = _gpu_encode(pcm, bps, rate, level, exhaustive=ex, prec_search=ps, max_batch=64)
        o = po.oracle_encode(pcm, bps, rate, level, exhaustive=ex, prec_search=ps)
        assert np.array_equal(fb, o["frame_bytes"]), (fam, bps, level, mode)
        assert data == o["data"], (fam, bps, level, mode)


@pytest.mark.parametrize("order", [1, 2, 4, 5, 7, 9, 10, 11, 13, 14, 15, 16, 17, 24, 31, 32])
def test_every_lpc_order_class(order):
    """one max_lpc_order from every autocorrelation-routine / FIR-width class (the MAXORD 8 / 12 / 16 kernel instances)"""
    for bps in (16, 24):
        pcm = signals.music(4096 * 2 + 501, 2, bps, seed=order)
        for level in (5, 8):
            data, fb = _gpu_encode(pcm, bps, 96000, level, max_lpc_order=order, max_batch=16)
            o = po.oracle_encode(pcm, bps, 96000, level, max_lpc_order=order)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (order, bps, level)


@pytest.mark.parametrize("blocksize", [192, 256, 576, 1000, 1024, 2304, 4000, 4608, 8192, 16384])
def test_block_sizes(blocksize):
    """block sizes on and off the fast paths (multiples of 64 / 16 or not, one or several 4096-sample passes, Rice
    partitions of every size), incl. a short last block"""
    for bps, level in ((16, 2), (16, 8), (24, 8)):
        pcm = signals.music(blocksize * 3 + blocksize // 3 + 7, 2, bps, seed=blocksize % 97)
        data, fb = _gpu_encode(pcm, bps, 96000, level, blocksize=blocksize, max_batch=8)
        o = po.oracle_encode(pcm, bps, 96000, level, blocksize=blocksize)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (blocksize, bps, level)


@pytest.mark.parametrize("po_range", [(0, 0), (0, 2), (3, 3), (2, 6), (0, 8), (8, 8)])
def test_partition_order_ranges(po_range):
    lo, hi = po_range
    pcm = signals.mixed(4096 * 3 + 99, 2, 16)
    for level in (2, 8):
        data, fb = _gpu_encode(pcm, 16, 44100, level, min_partition_order=lo, max_partition_order=hi, max_batch=8)
        o = po.oracle_encode(pcm, 16, 44100, level, min_po=lo, max_po=hi)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (po_range, level)


TINY_ORDERS = (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32)
DISABLES = ((0, 0, 0), (1, 0, 1), (1, 1, 1))


@pytest.mark.parametrize("blocksize", [16, 17, 19, 24, 31, 32, 33])
def test_tiny_blocks_all_orders(blocksize):
    """test/test_streams.sh:221-239: 8-bit mono noise, -8 -p -e -l <order> --lax --blocksize=<16..33>, with the
    constant / fixed / verbatim subframes switched off in turn, and the same with subdivide_tukey(32)"""
    pcm = signals.white(blocksize * 5 + 7, 1, 8, seed=blocksize)
    for order in TINY_ORDERS:
        if order > blocksize:
            continue
        for kw in [dict(disable=d) for d in DISABLES] + [dict(apodization="subdivide_tukey(32)")]:
            okw = dict(kw)
            if "apodization" in okw:
                okw["apod"] = ("subdivide_tukey", 32)
                del okw["apodization"]
            data, fb = _gpu_encode(pcm, 8, 44100, 8, blocksize=blocksize, max_lpc_order=order, exhaustive=1, prec_search=1,
                                   streamable_subset=0, max_batch=8, **kw)
            o = po.oracle_encode(pcm, 8, 44100, 8, blocksize=blocksize, max_lpc_order=order, exhaustive=1, prec_search=1, **okw)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (blocksize, order, kw)


@pytest.mark.parametrize("order", [16, 32])
def test_high_orders_with_searches(order):
    """test/test_streams.sh:181-219: -0 -l 16|32 --lax -m -e -p on sines and full-scale streams, 8 / 16 / 24 bits"""
    for bps in (8, 16, 24):
        for fam, ch in (("sine", 1), ("music", 2), ("square", 2)):
            pcm = signals.FAMILIES[fam](1152 * 2 + 301, ch, bps)
            data, fb = _gpu_encode(pcm, bps, 44100, 0, max_lpc_order=order, exhaustive=1, prec_search=1, mid_side=1,
                                   loose_mid_side=0, streamable_subset=0, max_batch=8)
            o = po.oracle_encode(pcm, bps, 44100, 0, max_lpc_order=order, exhaustive=1, prec_search=1, mid_side=1, loose=0)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (order, bps, fam)


@pytest.mark.parametrize("kw", [dict(blocksize=1000), dict(blocksize=4096, max_lpc_order=20, streamable_subset=0), dict(blocksize=576),
                                dict(blocksize=33, max_lpc_order=32, streamable_subset=0)],
                         ids=["b1000", "l20", "b576", "b33l32"])
def test_constant_and_silent_channels_off_the_fast_paths(kw):
    """CONSTANT subframes (a channel, or only the mid channel of a full-scale square pair) where the general kernels do the work"""
    for fam in ("square", "constant", "silence", "mixed"):
        pcm = signals.FAMILIES[fam](kw["blocksize"] * 3 + 77, 2, 16)
        data, fb = _gpu_encode(pcm, 16, 44100, 8, max_batch=8, **kw)
        okw = {k: v for k, v in kw.items() if k != "streamable_subset"}
        o = po.oracle_encode(pcm, 16, 44100, 8, **okw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (fam, kw)


@pytest.mark.parametrize("rate", [9, 90, 8000, 22050, 90000, 96000, 192000, 352800, 655350, 1048575])
def test_sample_rate_codes(rate):
    """test/test_streams.sh:241-250 frame-header variations: the sample-rate field (framing.c:289-329)"""
    pcm = signals.music(4096 + 300, 1, 16, seed=rate % 97)
    data, fb = _gpu_encode(pcm, 16, rate, 5, streamable_subset=0, max_batch=8)
    o = po.oracle_encode(pcm, 16, rate, 5)
    assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], rate


@pytest.mark.parametrize("bps", [25, 27, 28, 31, 32])
def test_wide_samples(bps):
    """more than 24 bits per sample: overflow-checked estimators and residuals, 64-bit fixed residuals, and at 32 bits the
    33-bit side channel (64-bit samples through prep / autocorrelation / evaluation / pack)"""
    for fam in ("music", "white", "sine", "square", "mixed", "wasted", "quiet", "constant", "silence"):
        for level in (0, 1, 2, 5, 8):
            for ch in (1, 2):
                pcm = signals.FAMILIES[fam](4096 + 1333, ch, bps)
                data, fb = _gpu_encode(pcm, bps, 96000, level, streamable_subset=0, max_batch=8)
                o = po.oracle_encode(pcm, bps, 96000, level)
                assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (bps, fam, level, ch)


@pytest.mark.parametrize("bps", [8, 16, 24, 32])
@pytest.mark.parametrize("pattern", range(1, 8))
def test_full_scale_deflection(bps, pattern):
    """test/test_streams.sh:188-193: fsd<bps>-0<pattern>, -0 -l 16 --lax -m -e -p; mono like the suite, and as an inverted pair"""
    for ch in (1, 2):
        pcm = signals.fsd(1152 * 2 + 100, ch, bps, pattern)
        kw = dict(max_lpc_order=16, exhaustive=1, prec_search=1, mid_side=1)
        data, fb = _gpu_encode(pcm, bps, 44100, 0, streamable_subset=0, loose_mid_side=0, max_batch=8, **kw)
        o = po.oracle_encode(pcm, bps, 44100, 0, loose=0, **kw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (bps, pattern, ch)


@pytest.mark.parametrize("tail", [1, 5, 6, 7, 9, 33, 1001, 4095])
def test_wide_samples_short_last_block(tail):
    """(n-4) % 4 != 0 in the four-lane estimators of the overflow-checked flavour"""
    for bps in (28, 32):
        for fam in ("music", "white"):
            pcm = signals.FAMILIES[fam](4096 + tail, 2, bps)
            for level, kw in ((5, {}), (8, dict(exhaustive=1))):
                data, fb = _gpu_encode(pcm, bps, 96000, level, streamable_subset=0, max_batch=8, **kw)
                o = po.oracle_encode(pcm, bps, 96000, level, **kw)
                assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (bps, fam, tail, level)


@pytest.mark.parametrize("kind", range(4))
def test_overflow_checked_residual_at_24_bits(kind):
    """smooth anti-phase pairs: candidates whose bound on the residual width exceeds 32 bits take the overflow-checked FIR
    (lpc.c:832) already at 24 bits per sample -- on the owner-layout evaluation and through pack2"""
    for bps in (24, 20):
        pcm = signals.slow(4096 * 2 + 77, 2, bps, kind)
        for level, kw in ((5, {}), (8, {}), (8, dict(exhaustive=1)), (5, dict(prec_search=1))):
            data, fb = _gpu_encode(pcm, bps, 96000, level, streamable_subset=0, max_batch=8, **kw)
            o = po.oracle_encode(pcm, bps, 96000, level, **kw)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (kind, bps, level, kw)


def _random_config(rng):
    """one legal encoder configuration + signal, drawn over the whole range the engine accepts"""
    bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24, 24, 32, int(rng.integers(4, 33))]))
    ch = int(rng.choice([1, 2, 2, 2, 3, 6, 8]))
    blocksize = int(rng.choice([16, 17, 64, 192, 576, 1000, 1024, 1152, 2048, 2304, 4096, 4096, 4096, 4608, 8192, 16384, int(rng.integers(16, 5000)),
                                int(rng.choice([16385, 32768, 65535, int(rng.integers(16385, 65536))]))]))
    lpc = int(rng.choice([0, 1, 4, 6, 8, 12, 12, 15, 16, 20, 32, int(rng.integers(0, 33))]))
    lpc = min(lpc, blocksize)
    kw = dict(blocksize=blocksize, max_lpc_order=lpc, streamable_subset=0)
    kw["max_partition_order"] = int(rng.integers(0, 9))
    kw["min_partition_order"] = int(rng.integers(0, kw["max_partition_order"] + 1))
    if ch == 2:
        kw["mid_side"] = int(rng.integers(0, 2))
        kw["loose_mid_side"] = int(rng.integers(0, 2)) if kw["mid_side"] else 0
    if lpc:
        kw["qlp_coeff_precision"] = int(rng.choice([0, 0, 5, 9, 12, 15]))
        if kw["qlp_coeff_precision"] == 0:
            del kw["qlp_coeff_precision"]
        kw["apodization"] = str(rng.choice(["tukey(0.5)", "subdivide_tukey(2)", "subdivide_tukey(3)", "hann;partial_tukey(2)", "welch;punchout_tukey(3/0.2)",
                                            "gauss(0.3)", "blackman;flattop", "subdivide_tukey(4/0.3)"]))
        kw["exhaustive"] = int(rng.random() < 0.25)
        kw["prec_search"] = int(rng.random() < 0.15)
    kw["limit_min_bitrate"] = int(rng.random() < 0.2)
    kw["disable"] = tuple(int(rng.random() < 0.15) for _ in range(3))
    fam = str(rng.choice(["music", "white", "sine", "mixed", "wasted", "square", "quiet", "constant", "silence"]))
    nframes = 1 + int(rng.integers(0, 3))
    n = blocksize * nframes + int(rng.integers(0, blocksize))
    n = max(n, 1)
    if blocksize >= 8192 and (kw.get("exhaustive") or kw.get("prec_search")):
        n = min(n, blocksize + 100)
    if blocksize > 16384:
        n = min(n, blocksize + 1000)
    rate = int(rng.choice([8000, 22050, 44100, 48000, 96000, 192000, 12345]))
    return fam, n, ch, bps, rate, kw


@pytest.mark.parametrize("seed", range(int(os.environ.get("FLACGPU_TEST_SEEDS", "200"))))
def test_random_configurations(seed, monkeypatch):
    """seeded sweep over the configuration space (block size, width, channels, orders, partition orders, apodizations,
    searches, disable switches, short last blocks): GPU == oracle driven by the same resolved settings, the device's own
    verify pass accepts every batch, and every engine starts from poisoned scratch memory (FLACGPU_POISON: a kernel that reads
    what no kernel wrote fails here)"""
    import flac_amd
    if monkeypatch is not None:
        monkeypatch.setenv("FLACGPU_POISON", "1")
    from oracle_from_settings import oracle_encode_settings
    rng = np.random.default_rng(1000 + seed)
    done = 0
    for _ in range(14):
        fam, n, ch, bps, rate, kw = _random_config(rng)
        pcm = signals.FAMILIES[fam](n, ch, bps)
        try:
            s = flac_amd.make_settings(ch, bps, rate, 5, **kw)
        except flac_amd.FlacGpuError:
            continue                                   # the reference's init would refuse this combination, too
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)          # nothing make_settings accepts is refused
        try:
            eng.set_verify(True)                       # every batch is decoded again on the device (the hinted pass where it applies)
            data, fb = eng.encode(pcm)
            v = eng.last_verify_result()
            in_range = int(pcm.min()) >= -(1 << (bps - 1)) and int(pcm.max()) < (1 << (bps - 1))
            # (a test signal that does not fit the stream's width -- `constant` is 1234 at any width -- encodes like the reference
            #  encodes it and need not decode back to itself: a predicted subframe carries it, a verbatim one truncates it)
            assert v.status == 0 if in_range else v.status in (0, 1), ("verify", v.status, v.frame_number, v.channel, v.sample, fam, n, ch, bps, rate, kw)
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (fam, n, ch, bps, rate, kw)
        done += 1
    assert done >= 6


@pytest.mark.parametrize("ch,bps,blocksize", [(6, 16, 16384), (8, 24, 16384), (8, 32, 8192), (3, 20, 16384)])
def test_frames_larger_than_the_lds(ch, bps, blocksize):
    """many channels x long blocks: the frame image is assembled in its HBM slot instead of the LDS"""
    for fam, level in (("music", 5), ("white", 8), ("mixed", 2)):
        pcm = signals.FAMILIES[fam](blocksize * 2 + 1234, ch, bps)
        data, fb = _gpu_encode(pcm, bps, 96000, level, blocksize=blocksize, streamable_subset=0, max_batch=4)
        o = po.oracle_encode(pcm, bps, 96000, level, blocksize=blocksize)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (ch, bps, blocksize, fam, level)


def test_longest_block_widest_search():
    """test/test_streams.sh:268: -b 16384 -m -r 8 -l 32 --lax -e -p"""
    for fam, ch in (("sine", 1), ("music", 2)):
        pcm = signals.FAMILIES[fam](16384 + 500, ch, 16)
        kw = dict(blocksize=16384, max_lpc_order=32, exhaustive=1, prec_search=1)
        data, fb = _gpu_encode(pcm, 16, 44100, 5, mid_side=1, loose_mid_side=0, max_partition_order=8, streamable_subset=0, max_batch=2, **kw)
        o = po.oracle_encode(pcm, 16, 44100, 5, mid_side=1, loose=0, max_po=8, **kw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (fam, ch)


@pytest.mark.parametrize("blocksize", [16385, 20000, 32768, 65535])
def test_blocks_longer_than_16384(blocksize):
    """test/test_streams.sh:243 (-b 65535): blocks that do not fit the LDS -- the general kernels read HBM instead"""
    for fam, ch, bps, level, kw in (("music", 2, 16, 5, {}), ("sine", 1, 16, 0, dict(max_lpc_order=32, exhaustive=1, mid_side=1)),
                                    ("mixed", 2, 24, 8, {}), ("white", 2, 32, 5, {}), ("wasted", 3, 16, 2, {})):
        pcm = signals.FAMILIES[fam](blocksize + 777, ch, bps)
        if ch != 2:
            kw = {k: v for k, v in kw.items() if k != "mid_side"}
        data, fb = _gpu_encode(pcm, bps, 44100, level, blocksize=blocksize, streamable_subset=0, max_batch=2, **kw)
        o = po.oracle_encode(pcm, bps, 44100, level, blocksize=blocksize, **kw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (blocksize, fam, ch, bps, level)


def test_many_apodizations_and_deep_subdivision():
    """-A lists up to FLAC__MAX_APODIZATION_FUNCTIONS (32) windows, and subdivide_tukey up to 32 parts (528 window jobs,
    1053 LPC analyses per subframe)"""
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    specs = [";".join(["hann", "welch", "tukey(0.3)", "gauss(0.2)", "blackman", "flattop", "nuttall", "bartlett"] * 4),
             "subdivide_tukey(12)", "subdivide_tukey(32)", "tukey(0.5);partial_tukey(4);punchout_tukey(5);subdivide_tukey(7)"]
    for spec in specs:
        for ch, bps, n in ((2, 16, 4096 + 321), (1, 24, 4096 * 2)):
            pcm = signals.music(n, ch, bps, seed=len(spec))
            s = flac_amd.make_settings(ch, bps, 44100, 8, apodization=spec, streamable_subset=0)
            eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=4)
            try:
                data, fb = eng.encode(pcm)
            finally:
                eng.close()
            o = oracle_encode_settings(pcm, s)
            assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (spec, ch, bps)


def test_parity_with_poisoned_scratch_memory(monkeypatch):
    """FLACGPU_POISON=1: an engine's scratch buffers start as 0xA5 garbage instead of the zeros fresh device memory happens to
    hold.  A kernel that reads what no kernel wrote (round 2 found one: taps 16..31 of the fixed-predictor records, read by the
    32-tap kernels, visible only when a process reused freed device memory) fails parity here, whatever ran before."""
    monkeypatch.setenv("FLACGPU_POISON", "1")
    for bs in (19, 33):
        test_tiny_blocks_all_orders(bs)
    for level in (0, 3, 5, 8):
        test_gpu_matches_oracle_per_frame(level)
    test_high_orders_with_searches(32)
    test_short_last_block(1)
    test_short_last_block(4095)
    test_wide_samples(32)
    test_wide_samples(25)
    test_limit_min_bitrate_and_channel_counts()
    test_wider_model_searches_match_oracle(8, (1, 1))
    test_frames_larger_than_the_lds(8, 24, 16384)
    test_blocks_longer_than_16384(65535)
    test_many_apodizations_and_deep_subdivision()
    for seed in range(6):
        test_random_configurations(seed, None)


def test_asynchronous_entry_matches_the_synchronous_one():
    """flacgpu_submit_batch_raw / flacgpu_collect: several batches in flight (their own device buffers, the input copies and the
    read-backs beside the kernels), collected in order -- the same bytes as one synchronous call per batch, a short last block and
    device verification included; a fifth submission is refused, a collect with nothing in flight as well"""
    import flac_amd
    from rawfmt import to_raw
    N, NF = 4096, 24
    pcm = signals.music(N * NF * 5 + 777, 2, 16, seed=21)
    eng = _engine(2, 16, 44100, 8, max_batch=NF)
    try:
        eng.set_verify(True)
        lib = eng.lib
        fmt = flac_amd.raw_format(16)
        cap = eng.max_output_bytes(NF)
        batches = []
        pos, first = 0, 0
        while pos < pcm.shape[0]:
            part = pcm[pos:pos + N * NF]
            nfr = (part.shape[0] + N - 1) // N
            tail = part.shape[0] - (nfr - 1) * N
            batches.append((np.ascontiguousarray(to_raw(part, 16)), nfr, first, 0 if tail == N else tail))
            pos += part.shape[0]
            first += nfr
        assert len(batches) == 6
        want = []
        for raw, nfr, first, tail in batches:
            data, fb = eng.encode_raw(raw, fmt, first_frame_number=first)
            want.append((data, fb.tolist()))
        assert lib.flacgpu_collect(eng.ctx) == -6                       # nothing in flight: FLACGPU_ERR_BAD_ARG
        outs = [np.zeros(cap, dtype=np.uint8) for _ in batches]
        fbs = [np.zeros(nfr, dtype=np.uint32) for _, nfr, _, _ in batches]
        tws = [eng._tail_windows(tail) if tail else None for _, _, _, tail in batches]
        sub = col = 0
        while col < len(batches):
            while sub < len(batches) and sub - col < 4:
                raw, nfr, first, tail = batches[sub]
                tw = tws[sub]
                r = lib.flacgpu_submit_batch_raw(eng.ctx, raw.ctypes.data, C.byref(fmt), nfr, first, tail or N, tw.ctypes.data if tw is not None else None,
                                                 outs[sub].ctypes.data, cap, fbs[sub].ctypes.data)
                assert r == 0, r
                sub += 1
            if sub - col == 4 and sub < len(batches):
                raw, nfr, first, tail = batches[sub]
                assert lib.flacgpu_submit_batch_raw(eng.ctx, raw.ctypes.data, C.byref(fmt), nfr, first, tail or N, None, outs[sub].ctypes.data, cap, fbs[sub].ctypes.data) == -8   # BUSY
            assert lib.flacgpu_in_flight(eng.ctx) == sub - col
            total = lib.flacgpu_collect(eng.ctx)
            assert total == len(want[col][0]), (col, total)
            assert outs[col][:total].tobytes() == want[col][0] and fbs[col].tolist() == want[col][1]
            assert eng.last_verify_result().status == 0
            col += 1
    finally:
        eng.close()


def test_two_wavefronts_per_channel_variant():
    """FLACGPU_EVAL_WPC=2 (evalg_kernel<., 2>: two wavefronts share a channel's image and halve its candidates; opt-in, read once per
    process): a fresh interpreter encodes the -8 and -6 cases with it and must produce the oracle's bytes"""
    import subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, flac_amd, signals\n"
            "from oracle import pyoracle as po\n"
            "for level, seed in ((8, 3), (6, 4)):\n"
            "    pcm = signals.music(4096 * 9 + 123, 2, 16, seed=seed)\n"
            "    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, level), device=0, max_batch_frames=16)\n"
            "    data, fb = eng.encode(pcm); eng.close()\n"
            "    assert data == po.oracle_encode(pcm, 16, 44100, level)['data'], level\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLACGPU_EVAL_WPC="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("level", [0, 1, 2])
def test_one_kernel_path_and_what_it_leaves(level):
    """ff_kernel (-0 .. -2 on 16-bit stereo, 1152-sample blocks: one kernel per batch) takes a frame only when its partition sums
    stay inside the search's 32-bit arithmetic; with `-r 0` a lane's 18 samples may sum to 2^17 at most, which full-scale noise
    exceeds and music does not: a stream of both has frames written by that kernel next to frames it left to the three-kernel
    path, whose workgroups skip the marked ones.  Same bytes as the oracle, with the default partition orders too."""
    quiet, loud = signals.music(1152 * 7, 2, 16, seed=21), signals.white(1152 * 6 + 500, 2, 16, seed=22)
    pcm = np.concatenate([quiet[: 1152 * 3], loud[: 1152 * 4], quiet[1152 * 3:], loud[1152 * 4:]])
    for kw, okw in ((dict(min_partition_order=0, max_partition_order=0), dict(min_po=0, max_po=0)), ({}, {})):
        data, fb = _gpu_encode(pcm, 16, 44100, level, max_batch=8, **kw)
        o = po.oracle_encode(pcm, 16, 44100, level, **okw)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (level, kw)


def test_three_kernel_path_of_the_fast_presets():
    """FLACGPU_NO_FF=1 (read once per process): -0 .. -2 on 16-bit stereo through prep2_kernel<., 1152, DECIDE>, eval_list_kernel and
    pack2_kernel<., 64, 18>, the path ff_kernel falls back to -- a fresh interpreter must produce the oracle's bytes with it"""
    import subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, flac_amd, signals\n"
            "from oracle import pyoracle as po\n"
            "for level, seed in ((0, 5), (1, 6), (2, 7)):\n"
            "    pcm = signals.music(1152 * 21 + 77, 2, 16, seed=seed)\n"
            "    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, level), device=0, max_batch_frames=16)\n"
            "    data, fb = eng.encode(pcm); eng.close()\n"
            "    assert data == po.oracle_encode(pcm, 16, 44100, level)['data'], level\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLACGPU_NO_FF="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("level,nframes,bs,tail", [(0, 8400, 1152, 0), (2, 4300, 1152, 321), (5, 4400, 4096, 0), (8, 700, 4096, 1000)])
def test_fused_output_over_many_segments(level, nframes, bs, tail):
    """The fused output (flacgpu_kernels.hip, PackOut: frames written once, at their final place; lengths published early, segment
    totals of 64 frames, segment starts) on batches of more than 64 segments, where a frame's walk over the segments in front of
    it crosses rows and meets known starts -- ff_kernel at -0 / -2, pack2_kernel at -5 / -8, with and without a short last block
    behind the frames of nominal length.  A repeating signal with per-repetition gain keeps the synthesis cheap and no two frames
    alike; bytes and frame lengths equal the oracle's."""
    base = signals.music(bs * 50, 2, 16, seed=40 + level).astype(np.int64)
    n = nframes * bs + tail
    reps = -(-n // len(base))
    pcm = np.concatenate([(base * (1000 - 3 * (r % 200))) // 1000 for r in range(reps)])[:n].astype(np.int32)
    eng = _engine(2, 16, 44100, level, max_batch=nframes + 1)
    try:
        for _ in range(2):                       # twice: the second batch finds the first one's (stale) words in the state arrays
            data, fb = eng.encode(pcm)
    finally:
        eng.close()
    o = po.oracle_encode(pcm, 16, 44100, level)
    assert np.array_equal(fb, o["frame_bytes"])
    assert data == o["data"]


def test_two_kernel_compaction_path():
    """FLACGPU_NO_FUSED_COMPACT=1 (read once per process): every frame to its slot, scan_kernel + compact_kernel behind the pack
    kernels -- what sub-batches on several streams and the debug stamps still use; a fresh interpreter must produce the oracle's bytes"""
    import subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, flac_amd, signals\n"
            "from oracle import pyoracle as po\n"
            "for level, bs, seed in ((0, 1152, 5), (2, 1152, 6), (5, 4096, 7), (8, 4096, 8)):\n"
            "    pcm = signals.music(bs * 70 + 77, 2, 16, seed=seed)\n"
            "    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, level), device=0, max_batch_frames=128)\n"
            "    data, fb = eng.encode(pcm); eng.close()\n"
            "    assert data == po.oracle_encode(pcm, 16, 44100, level)['data'], level\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLACGPU_NO_FUSED_COMPACT="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def _fresh_interpreter_cases(env, cases="((0, 1152, 1500, 5), (2, 1152, 700, 6), (5, 4096, 400, 7), (8, 4096, 150, 8))"):
    import subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, flac_amd, signals\n"
            "from oracle import pyoracle as po\n"
            "for level, bs, nfr, seed in " + cases + ":\n"
            "    pcm = signals.music(bs * nfr + 77, 2, 16, seed=seed)\n"
            "    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, 16, 44100, level), device=0, max_batch_frames=nfr + 1)\n"
            "    data, fb = eng.encode(pcm); data2, fb2 = eng.encode(pcm); eng.close()\n"
            "    o = po.oracle_encode(pcm, 16, 44100, level)\n"
            "    assert data == o['data'] and data2 == o['data'], level\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_fused_output_when_frames_give_up_waiting():
    """FLACGPU_FUSED_SPIN_LIMIT=0 (read when an engine is created): a frame whose predecessors' lengths are not there at its first
    look goes to its slot and onto the list of fo_place_kernel, which places it behind the pack kernel -- the route that keeps the
    fused output independent of the order in which the chip starts workgroups.  Same bytes; -5 / -8 (pack2_kernel places its own
    frames) and -0 / -2 with FLACGPU_FF_LAG (ff_kernel places the frame of n frames ago), a short last block behind them."""
    _fresh_interpreter_cases({"FLACGPU_FUSED_SPIN_LIMIT": "0", "FLACGPU_FF_LAG": "256"})


@pytest.mark.parametrize("lag", [0, 64, 2048])
def test_ff_kernel_places_the_frames_of_some_frames_ago(lag):
    """FLACGPU_FF_LAG=n (opt-in, launch_ff): ff_kernel publishes its lengths and the wavefront of frame f copies frame f - n from its
    slot to its place; the last n frames (n = 0, or a batch of no more than n frames: all of them) are fo_place_kernel's."""
    _fresh_interpreter_cases({"FLACGPU_FF_LAG": str(lag)}, "((0, 1152, 5000, 5), (1, 1152, 900, 9), (2, 1152, 1500, 6))")


def test_autoc3_kernel_a_lane_per_subframe():
    """FLACGPU_AUTOC3=1 (flacgpu_autoc.hip: autoc3_kernel whenever it applies, not only from 2048 wavefronts up -- the batch sizes of
    the tests are far below): stereo with a mid/side search, the three routines it instantiates (lag 8: -l 6; lag 12: -5; lag 16:
    -8), whole and partial windows, block sizes whose job lengths leave 1..3 chain steps in the last tile, wasted bits in one and in
    all channels, a pure tone (the ill-conditioned case that pins the association order), 24-bit samples, frame counts that do not
    fill the last wavefront's sixteen frames.  Same bytes as the oracle."""
    import subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, flac_amd, signals\n"
            "from oracle import pyoracle as po\n"
            "ran = []\n"
            "def run(pcm, bps, level, ekw={}, okw={}):\n"
            "    eng = flac_amd.FrameEngine(flac_amd.make_settings(2, bps, 44100, level, **ekw), device=0, max_batch_frames=64)\n"
            "    data, fb = eng.encode(pcm); ran.append('autoc3_kernel' in eng.last_batch_kernels()); eng.close()\n"
            "    o = po.oracle_encode(pcm, bps, 44100, level, **okw)\n"
            "    assert data == o['data'], (level, ekw)\n"
            "for level in (5, 8):\n"
            "    for bs, nfr in ((4096, 37), (4608, 21), (1152, 50), (576, 19), (4000, 17)):\n"
            "        run(signals.music(bs * nfr + 77, 2, 16, seed=bs + level), 16, level, dict(blocksize=bs), dict(blocksize=bs))\n"
            "    run(signals.wasted(4096 * 18, 2, 16), 16, level)\n"
            "    w = signals.music(4096 * 18, 2, 16, seed=3); w[:, 0] = (w[:, 0] >> 2) << 2\n"
            "    run(w, 16, level)\n"
            "    run(signals.sine(4096 * 20, 2, 16, freq=1000.0), 16, level)\n"
            "    run(signals.slow(4096 * 18, 2, 24), 24, level)\n"
            "    c = signals.music(4096 * 18, 2, 16, seed=5); c[:, 1] = 1234; c[4096 * 9:, 0] = -7\n"
            "    run(c, 16, level)\n"
            "    run(signals.mixed(4096 * 35 + 99, 2, 16), 16, level)\n"
            "run(signals.music(4096 * 33, 2, 16, seed=11), 16, 8, dict(max_lpc_order=6), dict(max_lpc_order=6))\n"
            "run(signals.music(4096 * 33, 2, 16, seed=12), 16, 8, dict(apodization='subdivide_tukey(2)'), dict(apod=('subdivide_tukey', 2)))\n"
            "run(signals.music(4096 * 33, 2, 16, seed=14), 16, 5, dict(apodization='subdivide_tukey(5)'), dict(apod=('subdivide_tukey', 5)))\n"
            "run(signals.music(4096 * 33, 2, 16, seed=13), 16, 3, dict(mid_side=1, loose_mid_side=0), dict(mid_side=1, loose=0))\n"
            "assert sum(ran) >= len(ran) - 2, ran       # (every case but the 24-bit one, whose planes are 32-bit words ... still autoc3; tiny margins)\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    # FLACGPU_AUTOC2=1: the streaming kernels whatever the batch size -- without it launch_analyze gives these small batches to the
    # wavefront-per-job kernel and autoc3_kernel never runs (round 4's version of this test: found in round 5 with the kernel record)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FLACGPU_AUTOC3="1", FLACGPU_AUTOC2="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
