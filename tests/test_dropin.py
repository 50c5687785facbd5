"""Binary-level drop-in: flac_amd/lib/libFLAC.so.14 carries the SONAME of the reference's library and exports
its whole public surface -- the encoder is this project's, the decoder / metadata / format code is the reference's own,
compiled unmodified -- so an ALREADY LINKED client gets the GPU encoder by nothing more than the library search path.

CPU: SONAME, every public libFLAC symbol of the reference library is exported, decoding / metadata editing through it
works (the reference's flac tool, linked once against the reference's libFLAC.so.14, decodes a file and runs --test with
LD_LIBRARY_PATH pointing here: those paths never touch the encoder).
GPU: the same binary encodes with LD_LIBRARY_PATH=oracle/_ref/dropin (reference) and LD_LIBRARY_PATH=flac_amd/lib (this
project): identical files; and with LD_PRELOAD=libFLACgpu.so over the reference library (symbol interposition)."""
import os
import re
import subprocess

import numpy as np
import pytest

import signals
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "flac_amd", "lib")
REFD = os.path.join(ROOT, "oracle", "_ref", "dropin")
FLAC = os.path.join(REFD, "flac")
needs = pytest.mark.skipif(not (os.path.exists(FLAC) and os.path.exists(os.path.join(OURS, "libFLAC.so.14"))), reason="drop-in binaries not built on this box")


def _run(libdir, args, preload=None, **kw):
    env = dict(os.environ, LD_LIBRARY_PATH=libdir)
    if preload:
        env["LD_PRELOAD"] = preload
    return subprocess.run([FLAC] + args, env=env, capture_output=True, **kw)


def _wav(path, pcm, bps=16, rate=44100):
    import struct
    n, ch = pcm.shape
    data = pcm.astype("<i2").tobytes() if bps == 16 else b"".join(int(v).to_bytes(3, "little", signed=True) for v in pcm.reshape(-1))
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * bps // 8, ch * bps // 8, bps) + b"data" + struct.pack("<I", len(data)) + data)


@needs
def test_soname_and_public_surface():
    so = os.path.join(OURS, "libFLAC.so.14")
    dyn = subprocess.check_output(["readelf", "-d", so]).decode()
    assert "Library soname: [libFLAC.so.14]" in dyn
    ours = set(re.findall(r" [TDRBW] (FLAC_\w+)", subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()))
    # what the public headers of the reference declare (FLAC_API ...): all of it must be here
    hdr = ""
    inc = os.path.join(ROOT, "include")
    ref_inc = "/root/reference/include/FLAC"
    if not os.path.isdir(ref_inc):
        pytest.skip("reference headers not on this box")
    for h in os.listdir(ref_inc):
        hdr += open(os.path.join(ref_inc, h)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    api = set(re.findall(r"FLAC_API\s+[^;{]*?\b(FLAC__\w+)\s*\(", hdr)) | set(re.findall(r"extern\s+FLAC_API\s+[^;]*?\b(FLAC__\w+)\s*(?:\[\])?;", hdr))
    assert len(api) > 150
    refsyms = set(re.findall(r" [TDRB] (FLAC_\w+)", subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(REFD, "libFLAC.so.14")]).decode()))
    missing = sorted(s for s in api if s in refsyms and s not in ours)
    assert not missing, missing


@needs
def test_already_linked_flac_decodes_and_tests_through_this_library(tmp_path):
    """decode / --test / metaflac-like listing never enter the encoder: they run here without a GPU"""
    pcm = signals.music(4096 * 20 + 123, 2, 16, seed=4)
    wav, fl, out = str(tmp_path / "a.wav"), str(tmp_path / "a.flac"), str(tmp_path / "b.wav")
    _wav(wav, pcm)
    r = _run(REFD, ["-s", "-f", "-5", "-o", fl, wav])
    assert r.returncode == 0, r.stderr
    # which library was loaded: ask the loader
    env = dict(os.environ, LD_LIBRARY_PATH=OURS)
    ldd = subprocess.check_output(["ldd", FLAC], env=env).decode()
    assert os.path.join(OURS, "libFLAC.so.14") in ldd and "libflacgpu.so" in ldd
    r = _run(OURS, ["-s", "-t", fl])
    assert r.returncode == 0, r.stderr
    r = _run(OURS, ["-s", "-f", "-d", "-o", out, fl])
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(wav, "rb").read()


@needs
@pytest.mark.gpu
def test_already_linked_flac_encodes_the_same_files_with_either_library(tmp_path):
    pcm = signals.music(4096 * 60 + 999, 2, 16, seed=6)
    wav = str(tmp_path / "a.wav")
    _wav(wav, pcm)
    for args in (["-8"], ["-5"], ["-0"], ["-8", "-V"], ["-6", "--padding=4096", "-S", "10x"], ["-8", "-e"], ["-3", "-T", "TITLE=x"]):
        a, b, c = str(tmp_path / "ref.flac"), str(tmp_path / "gpu.flac"), str(tmp_path / "pre.flac")
        r = _run(REFD, ["-s", "-f"] + args + ["-o", a, wav])
        assert r.returncode == 0, r.stderr
        r = _run(OURS, ["-s", "-f"] + args + ["-o", b, wav])
        assert r.returncode == 0, r.stderr
        assert open(a, "rb").read() == open(b, "rb").read(), args
        # symbol interposition instead of a search path: the reference library stays loaded, the encoder comes from libFLACgpu.so
        r = _run(REFD, ["-s", "-f"] + args + ["-o", c, wav], preload=os.path.join(OURS, "libFLACgpu.so"))
        assert r.returncode == 0, r.stderr
        assert open(a, "rb").read() == open(c, "rb").read(), args
    # and the file the GPU library wrote decodes bit-exactly with the reference library
    out = str(tmp_path / "back.wav")
    r = _run(REFD, ["-s", "-f", "-d", "-o", out, b])
    assert r.returncode == 0 and open(out, "rb").read() == open(wav, "rb").read()


@needs
def test_ogg_capability_flag_describes_the_encoder_only(tmp_path):
    """INTEGRATION.md, "What the binary drop-in does not do": libogg is not under /root/reference, so the reference's decoder and
    metadata halves are compiled into the drop-in with FLAC__HAS_OGG 0 while the encoder (this project's, with its own paging) does
    write Ogg FLAC.  FLAC_API_SUPPORTS_OGG_FLAC = 1 is therefore true of the encoder only: the decoder's init_ogg_* answer
    UNSUPPORTED_CONTAINER (stream_decoder.c:383: what a libFLAC built without libogg answers), cleanly and before touching the file."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(OURS, "libFLAC.so.14"))
    assert ctypes.c_int.in_dll(lib, "FLAC_API_SUPPORTS_OGG_FLAC").value == 1
    lib.FLAC__stream_decoder_new.restype = ctypes.c_void_p
    lib.FLAC__stream_decoder_init_ogg_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.FLAC__stream_decoder_init_file.argtypes = lib.FLAC__stream_decoder_init_ogg_file.argtypes
    lib.FLAC__stream_decoder_delete.argtypes = [ctypes.c_void_p]
    WR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    ER = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
    wr, er = WR(lambda *a: 0), ER(lambda *a: None)
    p = tmp_path / "x.oga"
    p.write_bytes(b"OggS" + bytes(60))
    dec = lib.FLAC__stream_decoder_new()
    assert dec
    rc = lib.FLAC__stream_decoder_init_ogg_file(dec, str(p).encode(), ctypes.cast(wr, ctypes.c_void_p), None, ctypes.cast(er, ctypes.c_void_p), None)
    assert rc == 1                  # FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER
    # the decoder object is still usable for native FLAC
    rc = lib.FLAC__stream_decoder_init_file(dec, str(p).encode(), ctypes.cast(wr, ctypes.c_void_p), None, ctypes.cast(er, ctypes.c_void_p), None)
    assert rc == 0
    lib.FLAC__stream_decoder_delete(dec)
