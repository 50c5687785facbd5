"""-m gpu: the device stream decoder (flacgpu_decode_stream_device, flac_amd/csrc/flacgpu_stream_decode.hip) against the REFERENCE's
decoder (oracle/_ref/libFLAC_ref.so: FLAC__stream_decoder_process_until_end_of_stream on the same bytes, oracle/ref_shim.c) on streams
this engine did not write: files of the reference's own `flac` tool over presets, `-e -p -l 32 --lax`, block sizes 16..65535, 8..32 bits,
1..8 channels -- samples and verdict must agree; on the same files damaged (flipped bits, false sync codes, cut, spliced, zeroed):
the same error callbacks in the same order and the same samples, silence for missing frames included.  SURVEY.md 8f row 3
(stream_decoder.c:2321 frame_sync_, :2373 read_frame_, :2624 read_frame_header_, :3299 read_residual_partitioned_rice_).
Nothing here reads /root/reference: oracle/_ref travels to the GPU box prebuilt."""
import hashlib
import os

import numpy as np
import pytest

import signals
import stream_decode_util as U

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not U.have_ref(), reason="oracle/_ref not built")]


@pytest.fixture(scope="module")
def dec():
    import flac_amd
    d = flac_amd.StreamDecoder(0)
    yield d
    d.close()


def check(dec, stream, what, expect_pcm=None, allow_long_rice=True):
    ref = U.ref_decode(stream)
    got = dec.decode(stream)
    v = U.same_verdict(ref, got)
    if v and allow_long_rice and got["long_rice_codes"]:
        return "skipped"
    assert v is None, (what, v)
    if expect_pcm is not None:
        assert np.array_equal(got["pcm"], expect_pcm), what
    return got


CLEAN = [
    (2, 16, 44100, ["-8"]), (2, 16, 44100, ["-5"]), (2, 16, 44100, ["-0"]), (2, 16, 44100, ["-2"]), (2, 16, 44100, ["-3"]),
    (1, 16, 44100, ["-5"]), (2, 24, 96000, ["-8"]), (2, 8, 8000, ["-4"]), (2, 32, 48000, ["-5"]), (2, 32, 48000, ["-8", "-e"]),
    (6, 16, 48000, ["-6"]), (8, 24, 48000, ["-5"]), (3, 16, 44100, ["-7"]), (5, 16, 44100, ["-1"]), (7, 16, 44100, ["-5"]), (4, 24, 44100, ["-8"]),
    (2, 16, 44100, ["-8", "-e", "-p"]), (2, 16, 44100, ["--lax", "-l", "32", "-8"]), (2, 24, 96000, ["--lax", "-l", "32", "-e", "-p", "-5"]),
    (2, 16, 44100, ["--lax", "-b", "16", "-l", "0"]), (2, 16, 44100, ["--lax", "-b", "65535", "-5"]), (2, 16, 44100, ["--lax", "-b", "1000", "-8"]),
    (2, 16, 44100, ["-b", "192", "-l", "4"]), (2, 16, 12345, ["--lax", "-5"]), (2, 16, 44100, ["-5", "--no-padding", "--no-seektable"]),
    (2, 16, 44100, ["-8", "-r", "0,0"]), (2, 16, 44100, ["-8", "-r", "8"]), (1, 8, 8000, ["-8", "--lax", "-b", "33"]),
]


def make_pcm(kind, n, ch, bps, seed):
    rng = np.random.default_rng(seed)
    if kind == "music":
        return signals.music(n, ch, bps, seed=seed)
    if kind == "noise":
        return rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(n, ch)).astype(np.int32)
    if kind == "wasted":
        return (signals.music(n, ch, bps, seed=seed) >> 3) << 3
    pcm = np.zeros((n, ch), dtype=np.int32)
    pcm[n // 2:] = 7
    return pcm


@pytest.mark.parametrize("case", range(len(CLEAN)))
@pytest.mark.parametrize("kind", ["music", "noise", "wasted", "silence"])
def test_reference_written_files_decode_as_the_reference_decodes_them(dec, case, kind):
    import torch
    ch, bps, rate, args = CLEAN[case]
    rng = np.random.default_rng(100 + case)
    bs = int(args[args.index("-b") + 1]) if "-b" in args else 4096
    n = min(bs * 5 + int(rng.integers(1, max(2, bs))), 200000)
    pcm = make_pcm(kind, n, ch, bps, case)
    f = U.flac_encode_cli(pcm, bps, rate, args)
    got = check(dec, f, (ch, bps, rate, args, kind), expect_pcm=pcm, allow_long_rice=False)
    assert got["events"] == [] and got["redecoded"] == 0 and got["silence"] == 0
    # the MD5 of STREAMINFO over the decoded samples (what FLAC__stream_decoder_finish compares, stream_decoder.c:670-676)
    d_pcm = torch.from_numpy(np.ascontiguousarray(got["pcm"])).to("cuda:0")
    bytes_per = (bps + 7) // 8
    d_bytes = torch.empty(d_pcm.numel() * bytes_per, dtype=torch.uint8, device="cuda:0")
    dec.pack_samples(d_pcm.data_ptr(), d_pcm.numel(), bps, d_bytes.data_ptr())
    torch.cuda.synchronize()
    assert hashlib.md5(d_bytes.cpu().numpy().tobytes()).digest() == got["md5"]


def damage(rng, f, first):
    b = bytearray(f)
    n = len(b)
    kind = int(rng.integers(0, 12))

    def pos():
        return int(rng.integers(first, n))
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            b[pos()] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        p = pos(); b[p:p + 2] = b'\xff\xf8'
    elif kind == 2:
        b = b[:pos()]
    elif kind == 3:
        p = pos(); q = min(n, p + int(rng.integers(1, 3000))); del b[p:q]
    elif kind == 4:
        p = pos(); b[p:p] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8))
    elif kind == 5:
        p = pos(); q = min(n, p + int(rng.integers(1, 64))); b[p:q] = b'\xff' * (q - p)
    elif kind == 6:
        p = pos(); q = min(n, p + int(rng.integers(1, 5000))); b[p:q] = bytes(q - p)
    elif kind == 7:
        p = pos(); b[p:p + 4] = bytes([0xff, 0xf8 | int(rng.integers(0, 2)), int(rng.integers(0, 256)), int(rng.integers(0, 256))])
    elif kind == 8:
        p = pos(); q = min(n, p + int(rng.integers(100, 20000))); b[q:q] = b[p:q]
    elif kind == 9:
        p = pos(); q = min(n, p + int(rng.integers(1, 100))); b[p:q] = bytes(rng.integers(0, 256, size=q - p, dtype=np.uint8))
    elif kind == 10:
        p = pos(); b = b[:first] + b[p:]
    else:
        for _ in range(int(rng.integers(2, 10))):
            p = pos(); b[p:p + 2] = bytes([0xff, 0xf8 + int(rng.integers(0, 2))])
    return bytes(b), kind


@pytest.mark.parametrize("seed", range(int(os.environ.get("FLACGPU_SD_SEEDS", "40"))))
def test_damaged_streams_give_the_references_errors_and_samples(dec, seed):
    rng = np.random.default_rng(5000 + seed)
    ch, bps, rate, args = CLEAN[int(rng.integers(0, len(CLEAN)))]
    bs = int(args[args.index("-b") + 1]) if "-b" in args else int(rng.choice([1152, 4096]))
    n = min(bs * int(rng.integers(2, 9)) + int(rng.integers(0, 1000)), 150000)
    pcm = make_pcm("music" if rng.random() < 0.7 else "noise", n, ch, bps, seed)
    f = U.flac_encode_cli(pcm, bps, rate, args)
    first = U.probe(f)[2]
    skipped = 0
    for d in range(8):
        g, kind = damage(rng, f, first)
        if check(dec, g, (seed, d, kind, ch, bps, rate, args)) == "skipped":
            skipped += 1
    assert skipped <= 4


@pytest.mark.parametrize("name", ["test_escape_coded_partitions_and_rice2", "test_variable_block_sizes_and_every_header_code", "test_reserved_and_broken_headers",
                                  "test_frames_missing_silence_and_its_caps", "test_values_that_overflow_and_32_bit_wrap_around",
                                  "test_orders_up_to_32_precisions_shifts_wasted_bits", "test_a_frame_hidden_in_verbatim_data_and_false_syncs_with_good_headers"])
def test_hand_built_streams_on_the_device(dec, name, monkeypatch):
    """The hand-built streams of tests/test_stream_decode_cpu.py (escape codes, RICE2, sample numbers and changing block sizes, every
    header code, streams without STREAMINFO, missing frames and the caps on their silence, overflowing values, a frame hidden in
    verbatim data, truncation at every byte) through the device decoder instead of the host build of its lane code."""
    import test_stream_decode_cpu as T

    def device_decode(stream, **kw):
        r = dec.decode(stream)
        r["retries"] = 1
        return r
    monkeypatch.setattr(U, "pin_decode", device_decode)
    getattr(T, name)()


GEN = os.path.join(U.ROOT, "oracle", "_ref", "test_streams")


@pytest.mark.skipif(not os.path.exists(GEN), reason="oracle/_ref/test_streams not built")
def test_the_reference_generators_streams_round_trip_through_the_device_decoder(dec, tmp_path):
    """The files of the reference's own stream generator (src/test_streams/main.c: sines, full-scale deflection, the rt-* files of
    1 / 111 / 4777 samples at 8..32 bits and 1..4+ channels, `wacky` headers) encoded by the reference's `flac` with the options of
    its test script (test/test_streams.sh:178-219) and decoded on the device: samples and verdict as the reference's decoder has them."""
    import subprocess
    subprocess.run([GEN], cwd=str(tmp_path), check=True, capture_output=True, timeout=300)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(U.FLAC_REF) + ":" + env.get("LD_LIBRARY_PATH", "")
    wavs = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".wav"))
    raws = [("sine16-%02d.raw" % k, 1 if k < 10 else 2, 16) for k in (0, 4, 3, 10, 14, 19)] + [("sine24-%02d.raw" % k, 1 if k < 10 else 2, 24) for k in (1, 12, 19)] + \
           [("sine32-%02d.raw" % k, 1 if k < 10 else 2, 32) for k in (3, 15)] + [("sine8-%02d.raw" % k, 1 if k < 10 else 2, 8) for k in (2, 16)] + \
           [("fsd%d-%02d.raw" % (b, k), 1, b) for b in (8, 16, 24, 32) for k in (1, 4, 7)]
    done = 0
    opts = [["-5"], ["-0", "-l", "16", "--lax", "-m", "-e", "-p"], ["-8", "-b", "1152"], ["--lax", "-l", "32", "-b", "4608"]]
    for k, w in enumerate(wavs):
        out = os.path.join(str(tmp_path), "o.flac")
        r = subprocess.run([U.FLAC_REF, "--silent", "-f", "-o", out] + opts[k % len(opts)] + [os.path.join(str(tmp_path), w)], env=env, capture_output=True)
        if r.returncode != 0:
            continue                                     # (the wacky files the tool itself refuses)
        check(dec, open(out, "rb").read(), w, allow_long_rice=False)
        done += 1
    for k, (name, ch, bps) in enumerate(raws):
        out = os.path.join(str(tmp_path), "o.flac")
        subprocess.check_call([U.FLAC_REF, "--silent", "-f", "-o", out, "--force-raw-format", "--endian=little", "--sign=signed", "--channels=%d" % ch, "--bps=%d" % bps,
                               "--sample-rate=44100"] + opts[k % len(opts)] + [os.path.join(str(tmp_path), name)], env=env)
        got = check(dec, open(out, "rb").read(), name, allow_long_rice=False)
        assert got["events"] == []
        done += 1
    assert done >= 40


def test_many_streams_decoded_and_their_md5_taken_on_the_device(dec):
    """`flac -t` over a directory, on the device end to end: every file decoded (flacgpu_decode_stream_device), its samples narrowed to
    the bytes the MD5 of STREAMINFO is over (flacgpu_pack_samples_device) and all digests taken by one launch, a lane per stream
    (flacgpu_md5_many_device) -- each equal to the file's STREAMINFO (what FLAC__stream_decoder_finish checks, stream_decoder.c:670-676)."""
    import torch
    from flac_amd.engine import md5_many_device
    files = []
    for k, (ch, bps, rate, args) in enumerate(CLEAN[:12]):
        pcm = make_pcm("music", 4096 * 3 + 17 * k, ch, bps, 40 + k)
        files.append((U.flac_encode_cli(pcm, bps, rate, args), pcm, bps))
    packed, offsets, lengths, want = [], [], [], []
    total = sum(p.size * ((b + 7) // 8) for _, p, b in files)
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device="cuda:0")
    off = 0
    for f, pcm, bps in files:
        got = dec.decode(f)
        assert got["events"] == [] and np.array_equal(got["pcm"], pcm)
        d_pcm = torch.from_numpy(np.ascontiguousarray(got["pcm"])).to("cuda:0")
        nb = d_pcm.numel() * ((bps + 7) // 8)
        dec.pack_samples(d_pcm.data_ptr(), d_pcm.numel(), bps, d_bytes.data_ptr() + off)
        torch.cuda.synchronize()
        offsets.append(off); lengths.append(nb); want.append(got["md5"])
        off += nb
    digests = md5_many_device(d_bytes.data_ptr(), offsets, lengths, device=0)
    assert digests == want


def test_flac_t_of_a_directory_on_the_device_agrees_with_the_references_tool(tmp_path):
    """python -m flac_amd.flactest (decode + MD5 of every file on the device) against `flac -t` of the reference's tool file by file:
    good files of several formats, a file with a flipped bit in its audio, one cut short, one whose STREAMINFO carries another MD5."""
    import subprocess
    import sys
    files, expect = [], []
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(U.FLAC_REF) + ":" + env.get("LD_LIBRARY_PATH", "")
    for k, (ch, bps, rate, args) in enumerate(CLEAN[:10]):
        pcm = make_pcm("music", 4096 * 4 + 33 * k, ch, bps, 70 + k)
        f = bytearray(U.flac_encode_cli(pcm, bps, rate, args))
        first = U.probe(bytes(f))[2]
        if k == 3:
            f[first + (len(f) - first) // 2] ^= 0x10
        elif k == 5:
            f = f[:first + (len(f) - first) * 2 // 3]
        elif k == 7:
            f[26 + 4] ^= 0xff                                   # a byte of STREAMINFO's MD5 (4 + 4 + 18 bytes in front of it)
        path = os.path.join(str(tmp_path), "f%02d.flac" % k)
        open(path, "wb").write(bytes(f))
        files.append(path)
        r = subprocess.run([U.FLAC_REF, "-t", "--silent", path], env=env, capture_output=True)
        expect.append(r.returncode == 0)
    assert expect.count(False) == 3
    r = subprocess.run([sys.executable, "-m", "flac_amd.flactest", "--json"] + files, capture_output=True, text=True, cwd=U.ROOT)
    import json
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert [f["ok"] for f in got["files"]] == expect, [(f["path"][-8:], f["ok"], f["errors"], f["md5"]) for f in got["files"]]
    assert r.returncode == 1
    assert got["files"][7]["md5"] == "mismatch" and got["files"][7]["errors"] == []
