"""The libFLAC stream-encoder API exported by libFLACgpu.so.

CPU part: every declared entry point is exported, settings/getters and init-time validation answer exactly like
the unmodified reference (oracle/_ref/libFLAC_ref.so) for the same calls, and without a GPU init refuses loudly.
GPU part (-m gpu): the same client session run against both libraries yields byte-identical .flac FILES --
STREAMINFO (MD5, total samples, min/max frame size), VORBIS_COMMENT, client metadata, seek table, frames."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import flac_api as fa
import signals
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/libFLAC_ref.so not built on this box")


def _declared(header):
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(FLAC__stream_encoder_[a-z0-9_A-Z]+)\s*\(", text)))


def test_every_declared_entry_point_is_exported():
    names = _declared(os.path.join(ROOT, "include", "FLACgpu_stream_encoder.h"))
    assert len(names) >= 60
    lib = C.CDLL(fa.GPU_SO)
    for n in names:
        assert hasattr(lib, n), n
    for table in ("FLAC__StreamEncoderStateString", "FLAC__StreamEncoderInitStatusString", "FLAC__StreamEncoderReadStatusString",
                  "FLAC__StreamEncoderWriteStatusString", "FLAC__StreamEncoderSeekStatusString", "FLAC__StreamEncoderTellStatusString",
                  "FLAC__VENDOR_STRING"):
        assert hasattr(lib, table), table


@needs_ref
def test_exports_cover_the_reference_encoder_surface():
    """every FLAC__stream_encoder_* symbol the reference library exports exists here too"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", fa.REF_SO]).decode()
    ref_syms = sorted(set(re.findall(r" [TDRB] (FLAC__stream_encoder_\w+|FLAC__StreamEncoder\w+String)", out)))
    assert len(ref_syms) >= 60
    lib = C.CDLL(fa.GPU_SO)
    missing = [s for s in ref_syms if not hasattr(lib, s)]
    assert not missing, missing


@needs_ref
def test_string_tables_match_reference():
    g, r = C.CDLL(fa.GPU_SO), C.CDLL(fa.REF_SO)
    for table, n in (("FLAC__StreamEncoderStateString", 9), ("FLAC__StreamEncoderInitStatusString", 14), ("FLAC__StreamEncoderReadStatusString", 4),
                     ("FLAC__StreamEncoderWriteStatusString", 2), ("FLAC__StreamEncoderSeekStatusString", 3), ("FLAC__StreamEncoderTellStatusString", 3)):
        a = (C.c_char_p * n).in_dll(g, table)
        b = (C.c_char_p * n).in_dll(r, table)
        assert list(a) == list(b)
    assert C.c_char_p.in_dll(g, "FLAC__VENDOR_STRING").value == C.c_char_p.in_dll(r, "FLAC__VENDOR_STRING").value


GETTERS = ("get_state", "get_verify", "get_streamable_subset", "get_channels", "get_bits_per_sample", "get_sample_rate", "get_blocksize",
           "get_do_mid_side_stereo", "get_loose_mid_side_stereo", "get_max_lpc_order", "get_qlp_coeff_precision",
           "get_do_qlp_coeff_prec_search", "get_do_exhaustive_model_search", "get_min_residual_partition_order",
           "get_max_residual_partition_order", "get_num_threads", "get_total_samples_estimate", "get_limit_min_bitrate",
           "get_verify_decoder_state")


def _snapshot(lib, e):
    return {g: getattr(lib, "FLAC__stream_encoder_" + g)(e) for g in GETTERS}


@needs_ref
def test_defaults_presets_and_setters_read_back_like_the_reference():
    g, r = fa.lib_for("gpu"), fa.lib_for("ref")
    eg, er = g.FLAC__stream_encoder_new(), r.FLAC__stream_encoder_new()
    try:
        assert _snapshot(g, eg) == _snapshot(r, er)                      # set_defaults_
        for level in list(range(9)) + [12]:
            assert g.FLAC__stream_encoder_set_compression_level(eg, level) == r.FLAC__stream_encoder_set_compression_level(er, level)
            assert _snapshot(g, eg) == _snapshot(r, er), level
        seq = (("set_channels", 6), ("set_bits_per_sample", 24), ("set_sample_rate", 96000), ("set_blocksize", 2048),
               ("set_max_lpc_order", 9), ("set_qlp_coeff_precision", 13), ("set_min_residual_partition_order", 2),
               ("set_max_residual_partition_order", 7), ("set_do_mid_side_stereo", 1), ("set_loose_mid_side_stereo", 1),
               ("set_streamable_subset", 0), ("set_limit_min_bitrate", 1), ("set_num_threads", 7), ("set_num_threads", 0),
               ("set_num_threads", 65), ("set_do_exhaustive_model_search", 1), ("set_do_qlp_coeff_prec_search", 1), ("set_verify", 1))
        for name, val in seq:
            assert getattr(g, "FLAC__stream_encoder_" + name)(eg, val) == getattr(r, "FLAC__stream_encoder_" + name)(er, val), name
            assert _snapshot(g, eg) == _snapshot(r, er), name
        assert g.FLAC__stream_encoder_set_total_samples_estimate(eg, 1 << 40) and r.FLAC__stream_encoder_set_total_samples_estimate(er, 1 << 40)
        assert _snapshot(g, eg) == _snapshot(r, er)
        assert g.FLAC__stream_encoder_get_resolved_state_string(eg) == r.FLAC__stream_encoder_get_resolved_state_string(er)
        # the pinned reference is built without libogg and refuses (0); this library carries its own Ogg layer (host/ogg.c)
        assert r.FLAC__stream_encoder_set_ogg_serial_number(er, 5) == 0 and g.FLAC__stream_encoder_set_ogg_serial_number(eg, 5) == 1
        assert g.FLAC__stream_encoder_finish(eg) == r.FLAC__stream_encoder_finish(er)      # legal on an uninitialised encoder
    finally:
        g.FLAC__stream_encoder_delete(eg); r.FLAC__stream_encoder_delete(er)


def _init_status(which, settings=(), metadata=None, callbacks="ok", channels=2, bps=16, rate=44100, ogg=False):
    lib = fa.lib_for(which)
    e = lib.FLAC__stream_encoder_new()
    try:
        lib.FLAC__stream_encoder_set_channels(e, channels)
        lib.FLAC__stream_encoder_set_bits_per_sample(e, bps)
        lib.FLAC__stream_encoder_set_sample_rate(e, rate)
        for name, val in settings:
            getattr(lib, "FLAC__stream_encoder_" + name)(e, val)
        keep = None
        if metadata is not None:
            keep = (C.POINTER(fa.StreamMetadata) * len(metadata))(*[C.pointer(m) for m in metadata])
            lib.FLAC__stream_encoder_set_metadata(e, keep, len(metadata))
        sink = fa.Sink()
        w, s, t, m = sink.callbacks()
        if callbacks == "nowrite":
            w = fa.WRITE_CB()
        elif callbacks == "seek-without-tell":
            t = fa.TELL_CB()
        if ogg:
            st = lib.FLAC__stream_encoder_init_ogg_stream(e, None, w, s, t, m, None)
        else:
            st = lib.FLAC__stream_encoder_init_stream(e, w, s, t, m, None)
        return st, lib.FLAC__stream_encoder_get_state(e)
    finally:
        lib.FLAC__stream_encoder_delete(e)


BAD_CONFIGS = [
    dict(channels=0), dict(channels=9), dict(bps=3), dict(bps=33), dict(rate=1048576),
    dict(settings=(("set_blocksize", 15),)), dict(settings=(("set_streamable_subset", 0), ("set_blocksize", 65536))),
    dict(settings=(("set_max_lpc_order", 33),)), dict(settings=(("set_blocksize", 16), ("set_max_lpc_order", 20), ("set_streamable_subset", 0))),
    dict(settings=(("set_qlp_coeff_precision", 4),)), dict(settings=(("set_qlp_coeff_precision", 16),)),
    dict(settings=(("set_blocksize", 8192),)),                 # not streamable at 44.1 kHz
    dict(settings=(("set_max_lpc_order", 14),)),               # not streamable at <= 48 kHz
    dict(bps=17), dict(rate=700000),
    dict(callbacks="nowrite"), dict(callbacks="seek-without-tell"),
]


@needs_ref
@pytest.mark.parametrize("cfg", BAD_CONFIGS, ids=lambda c: ",".join("%s=%s" % kv for kv in c.items()))
def test_init_rejects_bad_settings_with_the_reference_status(cfg):
    want = _init_status("ref", **cfg)
    assert want[0] != 0
    assert _init_status("gpu", **cfg) == want


@needs_ref
def test_init_rejects_illegal_metadata_like_the_reference():
    si = fa.StreamMetadata(); si.type = 0; si.length = 34
    cases = [
        [si],                                                          # a client STREAMINFO is illegal
        [fa.seektable([0, 1000]), fa.seektable([5])],                  # two seek tables
        [fa.seektable([1000, 1000])],                                  # not strictly ascending
        [fa.vorbis_comment([b"A=b"]), fa.vorbis_comment([])],          # two VORBIS_COMMENTs
        [fa.picture(1, b"image/png", b"", 16, 16, b"x")],              # file icon must be 32x32
        [fa.picture(1, b"image/png", b"", 32, 32, b"x"), fa.picture(1, b"image/png", b"", 32, 32, b"y")],
        [fa.picture(2, b"image/png", b"", 1, 1, b"x"), fa.picture(2, b"image/png", b"", 1, 1, b"y")],
        [fa.picture(3, b"image/\x01png", b"", 1, 1, b"x")],            # MIME type not printable ASCII
        [fa.picture(3, b"image/png", b"\xc0\x80", 1, 1, b"x")],        # overlong UTF-8 in the description
    ]
    for md in cases:
        want = _init_status("ref", metadata=md)
        assert want[0] == 12, want                                     # ..._INIT_STATUS_INVALID_METADATA
        assert _init_status("gpu", metadata=md) == want


def test_without_a_gpu_init_fails_loudly(capfd):
    """no CPU fallback: valid settings + no HIP device => ENCODER_ERROR and an explanation on stderr"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    st, state = _init_status("gpu")
    assert st == 1 and state != 0                                      # ENCODER_ERROR, state is an error state
    assert "cannot create the GPU frame engine" in capfd.readouterr().err


# ------------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _same_file(pcm, bps, rate, level, **kw):
    want, wsink = fa.encode("ref", pcm, bps, rate, level, **kw)
    got, gsink = fa.encode("gpu", pcm, bps, rate, level, **kw)
    assert len(got) == len(want)
    if got != want:
        first = next(i for i in range(len(got)) if got[i] != want[i])
        raise AssertionError("files differ first at byte %d of %d" % (first, len(got)))
    return got, gsink, wsink


@gpu
@needs_ref
@pytest.mark.parametrize("level", range(9))
def test_whole_file_identical_to_reference_all_presets(level):
    pcm = signals.music(4096 * 11 + 777, 2, 16, seed=level)
    data, gsink, wsink = _same_file(pcm, 16, 44100, level, chunk=1000)
    assert data[:4] == b"fLaC"
    # the write-callback protocol: same sequence of (bytes, samples, current_frame) once init is done; metadata first
    assert [c for c in gsink.calls] == [c for c in wsink.calls]
    # the STREAMINFO handed to the metadata callback is the same structure
    assert gsink.streaminfo[:72] == wsink.streaminfo[:72]


@gpu
@needs_ref
def test_md5_total_samples_and_framesizes_are_patched_into_streaminfo():
    pcm = signals.music(4096 * 5 + 100, 2, 16, seed=3)
    data, _, _ = _same_file(pcm, 16, 44100, 8)
    si = data[8:8 + 34]
    total = ((si[13] & 0x0F) << 32) | int.from_bytes(si[14:18], "big")
    assert total == len(pcm)
    raw = pcm.astype("<i2").tobytes()
    assert si[18:34] == hashlib.md5(raw).digest()
    assert int.from_bytes(si[4:7], "big") > 0 and int.from_bytes(si[7:10], "big") >= int.from_bytes(si[4:7], "big")


@gpu
@needs_ref
@pytest.mark.parametrize("n", [1, 15, 4095, 4096, 4097, 8192, 4096 * 3 + 1])
def test_stream_lengths_around_block_boundaries(n):
    """the one-sample overread: the final block (short or exactly full) is produced by finish()"""
    pcm = signals.music(n, 2, 16, seed=n)
    _same_file(pcm, 16, 44100, 5, chunk=[1, 7, 4096, 333])
    _same_file(pcm, 16, 44100, 8)


@gpu
@needs_ref
def test_planar_process_24bit_mono_and_multichannel():
    _same_file(signals.music(4096 * 3 + 50, 2, 24, seed=1), 24, 96000, 8, planar=True, chunk=999)
    _same_file(signals.music(4096 * 2 + 50, 1, 16, seed=2), 16, 44100, 5, planar=True)
    _same_file(signals.music(4096 * 2 + 9, 6, 16, seed=3), 16, 48000, 6, planar=True, chunk=4096)
    _same_file(signals.music(1152 * 3 + 9, 2, 8, seed=4), 8, 22050, 2, chunk=500)


@gpu
@needs_ref
def test_batches_smaller_than_the_stream(monkeypatch):
    """FLACGPU_BATCH_FRAMES=3: many engine calls per stream, identical file and callback sequence"""
    monkeypatch.setenv("FLACGPU_BATCH_FRAMES", "3")
    pcm = signals.mixed(4096 * 20 + 5, 2, 16)
    _, gsink, wsink = _same_file(pcm, 16, 44100, 8, chunk=2048)
    assert gsink.calls == wsink.calls


@gpu
@needs_ref
def test_client_metadata_seektable_and_padding():
    pcm = signals.music(4096 * 30 + 123, 2, 16, seed=9)
    md = lambda: [fa.seektable([0, 4096 * 3, 4096 * 3 + 5, 50000, 100000, 10 ** 7, 2 ** 64 - 1]),
                  fa.vorbis_comment([b"TITLE=x", b"ARTIST=\xc3\xa9"]), fa.application(b"abcd", b"payload bytes"),
                  fa.picture(3, b"image/png", b"cover", 1, 1, b"\x89PNG"), fa.padding(1000)]
    want, _ = fa.encode("ref", pcm, 16, 44100, 5, metadata=md(), chunk=5000)
    got, _ = fa.encode("gpu", pcm, 16, 44100, 5, metadata=md(), chunk=5000)
    assert got == want
    # non-seekable client: STREAMINFO stays unpatched in both
    want, _ = fa.encode("ref", pcm, 16, 44100, 5, metadata=md(), seekable=False)
    got, _ = fa.encode("gpu", pcm, 16, 44100, 5, metadata=md(), seekable=False)
    assert got == want


@gpu
@needs_ref
def test_init_file_progress_callback_and_total_samples_estimate(tmp_path):
    pcm = signals.music(4096 * 9 + 10, 2, 16, seed=5)
    prog = {"ref": [], "gpu": []}
    outs = {}
    for which in ("ref", "gpu"):
        def cb(enc, nbytes, nsamples, nframes, est, cd, which=which):
            prog[which].append((nbytes, nsamples, nframes, est))
        outs[which], _ = fa.encode(which, pcm, 16, 44100, 7, to_file=str(tmp_path / (which + ".flac")), progress=cb,
                                   total_samples_estimate=len(pcm), chunk=3000)
    assert outs["gpu"] == outs["ref"]
    assert prog["gpu"] == prog["ref"] and len(prog["gpu"]) == 10


@gpu
@needs_ref
def test_settings_through_the_api():
    pcm = signals.music(4096 * 6 + 17, 2, 16, seed=6)
    for settings in ((("set_blocksize", 1024),), (("set_apodization", b"hann;tukey(0.25);subdivide_tukey(2)"),),
                     (("set_max_lpc_order", 10), ("set_qlp_coeff_precision", 11)), (("set_do_mid_side_stereo", 0),),
                     (("set_max_residual_partition_order", 3), ("set_min_residual_partition_order", 2)),
                     (("set_limit_min_bitrate", 1),), (("set_do_md5", 0),), (("set_num_threads", 4),),
                     (("set_streamable_subset", 0), ("set_blocksize", 10000))):
        _same_file(pcm, 16, 44100, 6, settings=settings, chunk=4000)


@gpu
def test_out_of_range_sample_is_a_client_error_and_unsupported_features_are_refused():
    lib = fa.lib_for("gpu")
    pcm = signals.music(5000, 2, 16, seed=1)
    pcm[4000, 1] = 40000
    with pytest.raises(RuntimeError, match="CLIENT_ERROR"):
        fa.encode("gpu", pcm, 16, 44100, 5)
    assert lib is not None


@gpu
@needs_ref
def test_verify_through_the_api():
    """flac --verify: same file as the reference's (which verifies with its own decoder), verify state and stats readable"""
    pcm = signals.mixed(4096 * 5 + 200, 2, 16)
    for level in (0, 5, 8):
        _same_file(pcm, 16, 44100, level, settings=(("set_verify", 1),), chunk=5000)
    _same_file(signals.music(4096 * 2 + 77, 2, 24, seed=2), 24, 96000, 8, settings=(("set_verify", 1),))


@gpu
@needs_ref
def test_wider_model_searches_through_the_api():
    """flac -5e / -5p / -8ep: whole files identical to the reference's"""
    pcm = signals.music(4096 * 3 + 99, 2, 16, seed=8)
    for level, settings in ((5, (("set_do_exhaustive_model_search", 1),)), (5, (("set_do_qlp_coeff_prec_search", 1),)),
                            (8, (("set_do_exhaustive_model_search", 1), ("set_do_qlp_coeff_prec_search", 1)))):
        _same_file(pcm, 16, 44100, level, settings=settings, chunk=3000)


@gpu
@needs_ref
@pytest.mark.parametrize("spec", [b"partial_tukey(3/0.3/0.5)", b"punchout_tukey(2/0.2/0.4)", b"gauss(0.2);welch", b"subdivide_tukey(5)",
                                  b"blackman;hamming;flattop;nuttall", b"tukey(0.1);partial_tukey(2);punchout_tukey(3)", b"rectangle"])
def test_apodization_specs_through_the_api(spec):
    """window functions and their expansions (set_apodization, stream_encoder.c:1940-2070) through the whole stack"""
    pcm = signals.music(4096 * 3 + 517, 2, 16, seed=len(spec))
    _same_file(pcm, 16, 44100, 8, settings=(("set_apodization", spec),), chunk=4096)
    _same_file(pcm[:, :1], 16, 44100, 5, settings=(("set_apodization", spec), ("set_qlp_coeff_precision", 9)), chunk=777)


@gpu
@needs_ref
@pytest.mark.parametrize("precision", [5, 8, 14, 15])
def test_coefficient_precisions_through_the_api(precision):
    for bps, rate in ((16, 44100), (24, 96000)):
        pcm = signals.music(4096 * 2 + 123, 2, bps, seed=precision)
        _same_file(pcm, bps, rate, 8, settings=(("set_qlp_coeff_precision", precision),))


@gpu
@needs_ref
def test_wide_samples_and_high_orders_through_the_api():
    """32-bit input (33-bit side channel, MD5 over 4-byte samples, verify on), 28-bit, and -l 32: whole files identical"""
    lax = (("set_streamable_subset", 0),)
    _same_file(signals.music(4096 * 3 + 50, 2, 32, seed=1), 32, 96000, 8, settings=(("set_verify", 1),), chunk=999)
    _same_file(signals.fsd(4096 * 2 + 50, 2, 32, 3), 32, 48000, 5, settings=(("set_verify", 1),))
    _same_file(signals.white(4096 * 2 + 50, 2, 28), 28, 48000, 5, settings=lax + (("set_verify", 1),), planar=True)
    _same_file(signals.music(4096 * 2 + 50, 2, 16, seed=5), 16, 44100, 8, settings=lax + (("set_max_lpc_order", 32),))
    _same_file(signals.music(4096 * 2 + 50, 1, 24, seed=6), 24, 96000, 5, settings=lax + (("set_max_lpc_order", 17), ("set_do_exhaustive_model_search", 1)))


@gpu
@needs_ref
def test_several_encoders_at_once(monkeypatch):
    """four client threads, one encoder each (its own engine, HIP streams and worker thread), different settings:
    every file identical to the reference's"""
    import threading
    monkeypatch.setenv("FLACGPU_BATCH_FRAMES", "4")
    jobs = [(signals.music(4096 * 9 + 100, 2, 16, seed=1), 16, 44100, 8, ()), (signals.mixed(4096 * 7 + 5, 2, 16), 16, 44100, 5, (("set_verify", 1),)),
            (signals.music(4096 * 5 + 50, 2, 24, seed=2), 24, 96000, 8, ()), (signals.music(1152 * 11 + 9, 1, 16, seed=3), 16, 22050, 1, ())]
    want = [fa.encode("ref", p, b, r, l, settings=s)[0] for p, b, r, l, s in jobs]
    got = [None] * len(jobs)
    errs = []

    def run(i):
        try:
            p, b, r, l, s = jobs[i]
            got[i] = fa.encode("gpu", p, b, r, l, settings=s, chunk=1000 + 37 * i)[0]
        except Exception as e:          # noqa: BLE001
            errs.append((i, repr(e)))

    for _ in range(3):
        ths = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs
        for i in range(len(jobs)):
            assert got[i] == want[i], i


# ------------------------------------------------------------------------------------------------------------ Ogg FLAC
def _ogg_crc(data):
    """CRC-32 of an Ogg page, straight from RFC 3533: polynomial 0x04c11db7, MSB first, no reflection, init 0, no final xor"""
    crc = 0
    for byte in data:
        crc ^= byte << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04c11db7) & 0xffffffff if crc & 0x80000000 else (crc << 1) & 0xffffffff
    return crc


def _ogg_pages(data):
    """independent page parser: [(header_type, granule, serial, pageno, segments, body)]; checks capture pattern and CRC"""
    pages, pos = [], 0
    while pos < len(data):
        assert data[pos:pos + 4] == b"OggS" and data[pos + 4] == 0, pos
        htype = data[pos + 5]
        granule = int.from_bytes(data[pos + 6:pos + 14], "little", signed=True)
        serial = int.from_bytes(data[pos + 14:pos + 18], "little")
        pageno = int.from_bytes(data[pos + 18:pos + 22], "little")
        crc = int.from_bytes(data[pos + 22:pos + 26], "little")
        nseg = data[pos + 26]
        segs = list(data[pos + 27:pos + 27 + nseg])
        body = data[pos + 27 + nseg:pos + 27 + nseg + sum(segs)]
        page = bytearray(data[pos:pos + 27 + nseg + sum(segs)])
        page[22:26] = b"\0\0\0\0"
        assert _ogg_crc(page) == crc, ("page CRC", pageno)
        pages.append((htype, granule, serial, pageno, segs, body))
        pos += 27 + nseg + sum(segs)
    return pages


def _ogg_packets(pages):
    """packets in order, each with the index of the page it ends on"""
    packets, cur = [], b""
    for pi, (htype, granule, serial, pageno, segs, body) in enumerate(pages):
        assert bool(htype & 1) == bool(cur), "continued-packet flag"
        off = 0
        for s in segs:
            cur += body[off:off + s]
            off += s
            if s < 255:
                packets.append((cur, pi))
                cur = b""
    assert cur == b""
    return packets


@gpu
def test_ogg_flac_container(tmp_path):
    """FLAC__stream_encoder_init_ogg_stream / _file.  The reference here is built without libogg, so whole files are checked
    structurally (the paging itself is pinned against libogg-written streams in test_ogg_cpu.py): page structure and CRC with
    an independent parser, the mapping rules of ogg_encoder_aspect.c, packets == the native stream of the same encoder,
    granule positions, the STREAMINFO patched into the first page at finish."""
    for pcm, bps, rate, level, md, serial in ((signals.music(4096 * 23 + 777, 2, 16, seed=7), 16, 44100, 8, None, 0x1234567),
                                              (signals.white(4096 * 3 + 5, 2, 24), 24, 96000, 5, "vc+pad", 7),
                                              (signals.silence(4096 * 40, 1, 16), 16, 8000, 2, None, 0),
                                              (signals.music(100, 2, 16, seed=1), 16, 44100, 5, None, 1)):
        def mk():                # fresh blocks per session: the encoder sets is_last and fills the seek table in place
            return [fa.padding(100), fa.vorbis_comment([b"TITLE=x"]), fa.seektable([0, 4096])] if md else None
        meta = mk()
        native, _ = fa.encode("gpu", pcm, bps, rate, level, metadata=mk(), chunk=3000)
        settings = (("set_ogg_serial_number", serial),)
        ogg, sink = fa.encode("gpu", pcm, bps, rate, level, metadata=meta, chunk=3000, ogg=True, settings=settings)
        pages = _ogg_pages(ogg)
        assert [p[3] for p in pages] == list(range(len(pages)))                   # page sequence numbers
        assert all(p[2] == (serial & 0xffffffff) for p in pages)
        assert pages[0][0] & 2 and not any(p[0] & 2 for p in pages[1:])           # exactly one beginning-of-stream page
        assert pages[-1][0] & 4 and not any(p[0] & 4 for p in pages[:-1])         # the last page ends the stream
        packets = _ogg_packets(pages)
        # first packet: 0x7F "FLAC" 1.0, header-packet count, "fLaC", STREAMINFO -- alone on the first page
        first, first_page = packets[0]
        nmeta_client = len(meta) if meta else 0
        assert first[:13] == b"\x7fFLAC\x01\x00" + nmeta_client.to_bytes(2, "big") + b"fLaC" and len(first) == 13 + 38 and first_page == 0
        assert len(pages[0][5]) == len(first) and pages[0][1] == 0
        # the native stream, cut the same way: fLaC | STREAMINFO | other metadata blocks | frames
        blocks, pos, last = [], 4, False
        while not last:
            last = bool(native[pos] & 0x80)
            ln = int.from_bytes(native[pos + 1:pos + 4], "big")
            blocks.append(native[pos:pos + 4 + ln])
            pos += 4 + ln
        frames_native = native[pos:]
        assert first[13:] == blocks[0]                                            # STREAMINFO incl. MD5 / totals patched at finish
        if meta:
            # no seek table in Ogg FLAC and the VORBIS_COMMENT moves to the front (stream_encoder.c:831-857); the header count
            # is the number of blocks the client set, as in the reference (:2228)
            keep = [b for b in blocks[1:] if b[0] & 0x7f != 3]
            nblocks = len(keep)
            meta_packets = packets[1:1 + nblocks]
            assert [p[0] & 0x7f for p, _ in meta_packets] == [4, 1]
            assert sorted(p[1:] for p, _ in meta_packets) == sorted(b[1:] for b in keep)
            assert meta_packets[-1][0][0] & 0x80 and not meta_packets[0][0][0] & 0x80
        else:
            nblocks = len(blocks) - 1
            meta_packets = packets[1:1 + nblocks]
            assert [p for p, _ in meta_packets] == blocks[1:]
        assert [pi for _, pi in meta_packets] == list(range(1, 1 + nblocks))          # each metadata packet flushed to its own page
        # every metadata packet is flushed: the first audio packet starts a fresh page
        audio = packets[1 + nblocks:]
        assert b"".join(p for p, _ in audio) == frames_native
        assert all(p[:2] in (b"\xff\xf8", b"\xff\xf9") for p, _ in audio)          # one frame per packet
        if audio:
            assert audio[0][1] > meta_packets[-1][1] if meta_packets else audio[0][1] > 0
        # granule position of a page = samples up to the last packet ending on it; -1 if none ends there
        N = 1152 if level < 3 else 4096
        done = {}
        total = 0
        for i, (pk, pi) in enumerate(audio):
            total = min(len(pcm), (i + 1) * N)
            done[pi] = total
        for pi, pg in enumerate(pages):
            if pi in done:
                assert pg[1] == done[pi], pi
            elif pi > first_page and pi > (meta_packets[-1][1] if meta_packets else 0):
                assert pg[1] == -1, pi
        # pages close once more than 4096 body bytes are queued and four packets ended, or at 255 segments
        for pg in pages[:-1]:
            assert len(pg[4]) <= 255
        # the same through init_ogg_file
        path = str(tmp_path / ("t%d.oga" % serial))
        data, _ = fa.encode("gpu", pcm, bps, rate, level, metadata=mk(), chunk=3000, ogg=True, settings=settings, to_file=path)
        assert data == ogg
