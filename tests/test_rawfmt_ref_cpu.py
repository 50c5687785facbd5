"""CPU, where oracle/_ref holds the reference's own `flac` tool: the numpy restatement of format_input() (tests/rawfmt.py,
src/flac/encode.c:2352-2492) -- what the GPU staging kernel is compared with in the -m gpu tests -- pinned against files the
REFERENCE TOOL writes from the same raw bytes: every container width, both byte orders, both signs (--endian / --sign), and
the shifted containers of WAVEFORMATEXTENSIBLE (12 valid bits in 16, 20 in 24).  The tool encodes the raw bytes, decodes them
back to canonical little-endian signed samples, and those must be the integers rawfmt.format_input() says the encoder saw."""
import os
import struct
import subprocess

import numpy as np
import pytest

from rawfmt import format_input, to_raw

HERE = os.path.dirname(os.path.abspath(__file__))
CLI_REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "flac_cli_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI_REF), reason="oracle/_ref/flac_cli_ref not built (no /root/reference on this box)")


def _decode_to_int32(tmp_path, flac_file, bits, channels):
    out = str(tmp_path / "back.raw")
    r = subprocess.run([CLI_REF, "-d", "-s", "-f", "--force-raw-format", "--endian=little", "--sign=signed", "-o", out, flac_file], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(out, "rb").read()
    nb = (bits + 7) // 8
    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, channels, nb).astype(np.int64)
    v = np.zeros(b.shape[:2], dtype=np.int64)
    for k in range(nb):
        v |= b[:, :, k] << (8 * k)
    v = np.where(v >= (1 << (8 * nb - 1)), v - (1 << (8 * nb)), v)
    return v.astype(np.int32)


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
@pytest.mark.parametrize("be", [False, True])
@pytest.mark.parametrize("uns", [False, True])
def test_raw_containers_as_the_reference_tool_reads_them(tmp_path, bits, be, uns):
    rng = np.random.default_rng(7 * bits + 2 * be + uns)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    C = 2
    # arbitrary BYTES, not bytes derived from samples: what the restatement makes of them is the question
    raw = rng.integers(0, 256, size=600 * C * (bits // 8), dtype=np.uint8).tobytes()
    src = str(tmp_path / "in.raw")
    open(src, "wb").write(raw)
    enc = str(tmp_path / "x.flac")
    r = subprocess.run([CLI_REF, "-s", "-f", "-0", "--lax", "--force-raw-format", "--endian=%s" % ("big" if be else "little"), "--sign=%s" % ("unsigned" if uns else "signed"),
                        "--channels=%d" % C, "--bps=%d" % bits, "--sample-rate=44100", "-o", enc, src], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _decode_to_int32(tmp_path, enc, bits, C)
    got = format_input(raw, C, bits, big_endian=be, is_unsigned=uns)
    assert got.min() >= lo and got.max() <= hi
    assert np.array_equal(got, want)
    # and the inverse used by the GPU tests reproduces the bytes
    assert to_raw(got, bits, be, uns).tobytes() == raw


def _wavex(path, raw, channels, container_bits, valid_bits, rate):
    """WAVE_FORMAT_EXTENSIBLE with wValidBitsPerSample < the container: the tool encodes valid_bits-wide samples and shifts
    the container down (encode.c:1039-1046, :2479-2488)"""
    ba = channels * container_bits // 8
    fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, channels, rate, rate * ba, ba, container_bits, 22, valid_bits, 0x3 if channels == 2 else 0x4,
                      1, bytes([0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71]))
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)


@pytest.mark.parametrize("container,valid", [(16, 12), (24, 20), (32, 24)])
def test_shifted_containers_as_the_reference_tool_reads_them(tmp_path, container, valid):
    rng = np.random.default_rng(container + valid)
    shift = container - valid
    C = 2
    pcm = rng.integers(-(1 << (valid - 1)), 1 << (valid - 1), size=(500, C), dtype=np.int64).astype(np.int32)
    raw = to_raw(pcm, container, False, False, shift=shift).tobytes()          # left-justified in the container, low bits zero
    src = str(tmp_path / "in.wav")
    _wavex(src, raw, C, container, valid, 48000)
    enc = str(tmp_path / "x.flac")
    r = subprocess.run([CLI_REF, "-s", "-f", "-0", "-o", enc, src], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    # the stream holds `valid`-bit samples; decoded to WAVE the tool left-justifies them in the container again
    # (src/flac/decode.c: shift = 8 - bps % 8): the data chunk must be the bytes we started from, i.e. the samples the
    # encoder saw were (container value >> shift) -- what format_input() returns
    back = str(tmp_path / "back.wav")
    r = subprocess.run([CLI_REF, "-d", "-s", "-f", "-o", back, enc], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    w = open(back, "rb").read()
    at = w.index(b"data")
    n = struct.unpack("<I", w[at + 4:at + 8])[0]
    oc = (valid + 7) // 8 * 8                                           # the decoder's container: the next multiple of 8
    assert w[at + 8:at + 8 + n] == to_raw(pcm, oc, False, False, shift=oc - valid).tobytes()
    with open(enc, "rb") as f:
        si = f.read(42)[8:]                                            # STREAMINFO body
    assert ((si[12] & 1) << 4 | si[13] >> 4) + 1 == valid              # bits per sample of the stream
    got = format_input(raw, C, container, shift=shift)
    assert np.array_equal(got, pcm)
    # non-zero bits below the shift: the tool refuses the file, the restatement raises (the staging kernel reports it: test_gpu_parity)
    bad = bytearray(raw)
    bad[0] |= 1
    _wavex(src, bytes(bad), C, container, valid, 48000)
    r = subprocess.run([CLI_REF, "-s", "-f", "-0", "-o", enc, src], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    with pytest.raises(ValueError):
        format_input(bytes(bad), C, container, shift=shift)
