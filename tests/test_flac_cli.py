"""-m gpu: the reference's own `flac` command-line tool (src/flac, compiled unmodified into oracle/_ref by `make -C oracle cli`),
linked against libFLACgpu.so for the encoder -- the north star's "the flac CLI links against it unchanged" -- next to the same
tool linked against the reference library.  Same command line, same input file: the two .flac files must be identical
(STREAMINFO with MD5, seek table, padding, Vorbis comment and every frame)."""
import os
import subprocess
import wave

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
CLI_REF, CLI_GPU = os.path.join(REFDIR, "flac_cli_ref"), os.path.join(REFDIR, "flac_cli_gpu")
needs_cli = pytest.mark.skipif(not (os.path.exists(CLI_REF) and os.path.exists(CLI_GPU)), reason="oracle/_ref CLI binaries not built")


def _wav(path, pcm, bps, rate):
    w = wave.open(path, "wb")
    w.setnchannels(pcm.shape[1]); w.setsampwidth((bps + 7) // 8); w.setframerate(rate)
    if bps == 8:
        raw = (pcm + 128).astype(np.uint8).tobytes()
    elif bps == 16:
        raw = pcm.astype("<i2").tobytes()
    elif bps == 24:
        b = pcm.astype("<i4").tobytes()
        raw = b"".join(b[i:i + 3] for i in range(0, len(b), 4))
    else:
        raw = pcm.astype("<i4").tobytes()
    w.writeframes(raw)
    w.close()


def _run(cli, args, out):
    r = subprocess.run([cli] + args + ["-f", "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (cli, args, r.stderr[-2000:])
    with open(out, "rb") as f:
        return f.read()


CASES = [
    ("music", 2, 16, 44100, 3 * 44100 + 17, ["-8"]),
    ("music", 2, 16, 44100, 2 * 44100, ["-5", "--verify"]),
    ("mixed", 2, 16, 44100, 44100, ["-0"]),
    ("music", 1, 16, 22050, 30000, ["-3", "-S", "0.5s", "-P", "1000"]),
    ("music", 2, 24, 96000, 96000, ["-8", "-V", "-T", "TITLE=t", "-T", "ARTIST=a"]),
    ("white", 2, 8, 8000, 20000, ["-6", "--lax", "-b", "1000"]),
    ("music", 2, 16, 48000, 60000, ["-8", "-e", "-p", "--lax", "-l", "20"]),
    ("music", 2, 32, 48000, 40000, ["-7", "-V"]),
    ("square", 2, 16, 44100, 50000, ["-4", "-A", "tukey(0.25);partial_tukey(2)", "-r", "2,5"]),
    ("music", 6, 16, 48000, 30000, ["-5", "--channel-map=none"]),
]


@needs_cli
@pytest.mark.parametrize("case", CASES, ids=[" ".join([c[0], str(c[1]), str(c[2])] + c[5]) for c in CASES])
def test_unmodified_flac_tool_writes_the_same_file(tmp_path, case):
    fam, ch, bps, rate, n, args = case
    pcm = signals.FAMILIES[fam](n, ch, bps)
    wav = str(tmp_path / "in.wav")
    _wav(wav, pcm, bps, rate)
    want = _run(CLI_REF, args + [wav], str(tmp_path / "ref.flac"))
    got = _run(CLI_GPU, args + [wav], str(tmp_path / "gpu.flac"))
    assert len(got) == len(want)
    assert got == want


@needs_cli
def test_raw_input_and_decode_back(tmp_path):
    """raw PCM through the tool's own input staging, then the reference decoder reads our file back to the input"""
    pcm = signals.music(70000, 2, 16, seed=9)
    raw = str(tmp_path / "in.raw")
    pcm.astype("<i2").tofile(raw)
    args = ["-8", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=2", "--bps=16", "--sample-rate=44100", raw]
    want = _run(CLI_REF, args, str(tmp_path / "ref.flac"))
    got = _run(CLI_GPU, args, str(tmp_path / "gpu.flac"))
    assert got == want
    out = str(tmp_path / "back.raw")
    r = subprocess.run([CLI_REF, "-d", "--force-raw-format", "--endian=little", "--sign=signed", "-f", "-o", out, str(tmp_path / "gpu.flac")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert np.array_equal(np.fromfile(out, dtype="<i2").reshape(-1, 2), pcm.astype(np.int16))


CXX_REF, CXX_GPU = os.path.join(REFDIR, "cxx_encode_ref"), os.path.join(REFDIR, "cxx_encode_gpu")


@pytest.mark.skipif(not (os.path.exists(CXX_REF) and os.path.exists(CXX_GPU)), reason="oracle/_ref C++ clients not built")
@pytest.mark.parametrize("level", [0, 5, 8])
def test_cxx_wrapper_client(tmp_path, level):
    """the reference's C++ wrapper (libFLAC++ FLAC::Encoder::File, compiled unmodified) under a small client, on either library:
    same file; with init_ogg the client gets an Ogg FLAC file from this library (the pinned reference has no libogg)"""
    pcm = signals.music(4096 * 6 + 321, 2, 16, seed=level)
    raw = str(tmp_path / "in.raw")
    pcm.astype("<i2").tofile(raw)
    outs = []
    for exe in (CXX_REF, CXX_GPU):
        out = str(tmp_path / (os.path.basename(exe) + ".flac"))
        r = subprocess.run([exe, raw, out, str(level)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (exe, r.stderr)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    out = str(tmp_path / "o.oga")
    r = subprocess.run([CXX_GPU, raw, out, str(level), "ogg"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    data = open(out, "rb").read()
    assert data[:4] == b"OggS" and data[28:33] == b"\x7fFLAC" and int.from_bytes(data[14:18], "little") == 4711
    r = subprocess.run([CXX_REF, raw, out, str(level), "ogg"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0                     # UNSUPPORTED_CONTAINER: the reference build here has no libogg


SUITE_GPU = os.path.join(REFDIR, "api_suite_gpu")


@pytest.mark.skipif(not os.path.exists(SUITE_GPU), reason="oracle/_ref API suite not built")
def test_reference_encoder_api_unit_test_passes_on_this_library(tmp_path):
    """src/test_libFLAC/encoders.c -- the reference's own unit test of the FLAC__StreamEncoder API (every setter / getter, the four
    init layers, process, process_interleaved, finish, verify on), compiled unmodified and linked against libFLACgpu.so; this
    library exports FLAC_API_SUPPORTS_OGG_FLAC = 1, so the suite runs its Ogg FLAC half as well"""
    r = subprocess.run([SUITE_GPU], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("PASSED!") == 8 and "format: Ogg FLAC" in r.stdout and "ENCODER API SUITE PASSED" in r.stdout


SUITE_CXX_GPU = os.path.join(REFDIR, "api_suite_cxx_gpu")


@pytest.mark.skipif(not os.path.exists(SUITE_CXX_GPU), reason="oracle/_ref C++ API suite not built")
def test_reference_cxx_encoder_api_unit_test_passes_on_this_library(tmp_path):
    """src/test_libFLAC++/encoders.cpp on src/libFLAC++/stream_encoder.cpp -- the reference's unit test of FLAC::Encoder::Stream /
    ::File, both compiled unmodified, linked against libFLACgpu.so; native FLAC and Ogg FLAC halves"""
    r = subprocess.run([SUITE_CXX_GPU], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:]
    assert "C++ ENCODER API SUITE PASSED" in r.stdout and "Ogg FLAC" in r.stdout and "FAILED" not in r.stdout.replace("SUITE FAILED", "")


GEN = os.path.join(REFDIR, "test_streams")


@needs_cli
@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref/test_streams not built")
def test_baseline_config_1_through_the_tool(tmp_path):
    """BASELINE.json configs[0]: the reference's generator writes sine16-02/03/04.raw, and
    `flac -5 --force-raw-format --endian=little --sign=signed --channels=1 --bps=16 --sample-rate=44100` encodes them -- the same
    tool on either library, identical files (49 frames each); plus the generator's full-scale, wasted-bits and noise streams with
    the options of test/test_streams.sh"""
    subprocess.run([GEN], cwd=str(tmp_path), check=True, capture_output=True, timeout=300)
    fmt = ["--force-raw-format", "--endian=little", "--sign=signed", "--sample-rate=44100"]
    jobs = [("sine16-%s.raw" % nn, ["-5", "--channels=1", "--bps=16"]) for nn in ("02", "03", "04")]
    jobs += [("fsd%d-0%d.raw" % (bps, k), ["-0", "-l", "16", "--lax", "-m", "-e", "-p", "--channels=1", "--bps=%d" % bps]) for bps in (8, 16, 24, 32) for k in (1, 4, 7)]
    jobs += [("wbps16-01.raw", ["-0", "-l", "16", "--lax", "-m", "-e", "-p", "--channels=1", "--bps=16"]),
             ("sine24-13.raw", ["-0", "-l", "16", "--lax", "-m", "-e", "--channels=2", "--bps=24"]),
             ("noise.raw", ["-8", "--channels=2", "--bps=16"])]
    for name, args in jobs:
        src = str(tmp_path / name)
        want = _run(CLI_REF, args + fmt + [src], str(tmp_path / "ref.flac"))
        got = _run(CLI_GPU, args + fmt + [src], str(tmp_path / "gpu.flac"))
        assert got == want, name
