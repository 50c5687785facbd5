"""The host API layer's batch pipeline on a machine without a GPU: stream_encoder.c against tests/fake_engine (a stand-in
for libflacgpu.so that writes checkable records instead of FLAC frames -- test infrastructure, see its header).

What is pinned here: frames come out in stream order, each built from exactly its own samples (narrowing copy, with and
without the helper threads; batch boundaries and the overread sample; short last block), the STREAMINFO MD5 equals the MD5
of the little-endian sample bytes however the chain was cut into pieces (ahead of the submissions or not), total samples and
min/max frame size are those of the stream.  The same cases run against the real engine in tests/test_stream_encoder_api.py
(-m gpu), where the frames are FLAC and are compared with the reference's files."""
import hashlib
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE_DIR = os.path.join(HERE, "fake_engine")
FAKE_SO = os.path.join(FAKE_DIR, "libflacgpu.so")


def _build():
    src = os.path.join(FAKE_DIR, "fake_engine.c")
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", FAKE_SO])


def fnv1a(b):
    h = 2166136261
    for x in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ x) * 16777619) & 0xFFFFFFFF
    return h


CHILD = r'''
import os, sys, json, hashlib
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import flac_api
case = json.loads(sys.argv[1])
rng = np.random.default_rng(case["seed"])
bps, ch, n = case["bps"], case["channels"], case["samples"]
pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(n, ch), dtype=np.int64).astype(np.int32)
settings = [("set_blocksize", case["blocksize"]), ("set_do_md5", case["md5"])]
data, sink = flac_api.encode("gpu", pcm, bps, 44100, level=5, chunk=case["chunk"], planar=case["planar"], settings=settings)
sys.stdout.buffer.write(data)
'''


def run_case(case, env_extra):
    _build()
    import json
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, json.dumps(case)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return out.stdout


def expected_stream(case):
    rng = np.random.default_rng(case["seed"])
    bps, ch, n, N = case["bps"], case["channels"], case["samples"], case["blocksize"]
    pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(n, ch), dtype=np.int64).astype(np.int32)
    w = (bps + 7) // 8
    raw = pcm.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :w].tobytes()
    frames = []
    for f in range((n + N - 1) // N):
        a, b = f * N, min(n, (f + 1) * N)
        h = fnv1a(raw[a * ch * w:b * ch * w])
        frames.append(b"FK\0\0" + struct.pack("<III", f, b - a, h) + b"\xEE" * (f % 5))
    return raw, frames


def check(case, env_extra):
    data = run_case(case, env_extra)
    raw, frames = expected_stream(case)
    assert data[:4] == b"fLaC"
    assert data[4] == 0 and data[5:8] == b"\0\0\x22"                      # STREAMINFO, not last, 34 bytes
    si = data[8:42]
    N, n = case["blocksize"], case["samples"]
    min_bs, max_bs = struct.unpack(">HH", si[:4])
    assert (min_bs, max_bs) == (N, N)
    min_fs, max_fs = int.from_bytes(si[4:7], "big"), int.from_bytes(si[7:10], "big")
    assert (min_fs, max_fs) == (min(map(len, frames)), max(map(len, frames)))
    packed = int.from_bytes(si[10:18], "big")
    assert packed & ((1 << 36) - 1) == n
    assert (packed >> 36) & 31 == case["bps"] - 1 and (packed >> 41) & 7 == case["channels"] - 1
    assert si[18:34] == (hashlib.md5(raw).digest() if case["md5"] else bytes(16))
    # the VORBIS_COMMENT the layer adds, then the records
    assert data[42] == 0x84
    vlen = int.from_bytes(data[43:46], "big")
    body = data[46 + vlen:]
    assert body == b"".join(frames)


BASE = dict(seed=1, bps=16, channels=2, blocksize=256, samples=256 * 37 + 100, chunk=None, planar=False, md5=1)


@pytest.mark.parametrize("batch", [1, 2, 3, 7, 64])
@pytest.mark.parametrize("md5", [0, 1])
def test_batches_and_md5(batch, md5):
    check(dict(BASE, md5=md5), {"FLACGPU_BATCH_FRAMES": str(batch)})


@pytest.mark.parametrize("chunk", [1, 255, 256, 257, 1000, [3, 5000, 17]])
def test_chunkings(chunk):
    check(dict(BASE, chunk=chunk, samples=256 * 9 + 255), {"FLACGPU_BATCH_FRAMES": "4"})


@pytest.mark.parametrize("planar", [False, True])
@pytest.mark.parametrize("bps,channels", [(8, 1), (16, 2), (24, 2), (32, 3), (16, 6)])
def test_widths_and_layouts(planar, bps, channels):
    check(dict(BASE, bps=bps, channels=channels, planar=planar, chunk=777), {"FLACGPU_BATCH_FRAMES": "5"})


@pytest.mark.parametrize("threads", [1, 2, 4, 7])
@pytest.mark.parametrize("planar", [False, True])
def test_helper_threads_and_running_md5(threads, planar):
    """calls big enough for the helper threads (>= 2^18 values each) and batches long enough for the MD5 chain to run ahead of
    the submissions (pieces of 2^16 samples); the engine's pause lets the caller get ahead of the worker"""
    N = 4096
    case = dict(BASE, blocksize=N, samples=N * 150 + 1234, chunk=[N * 40 + 1, 200001, N * 64], planar=planar, seed=5)
    check(case, {"FLACGPU_BATCH_FRAMES": "48", "FLACGPU_STAGE_THREADS": str(threads), "FAKE_ENGINE_DELAY_US": "3000"})


def test_exact_multiple_and_tiny():
    check(dict(BASE, samples=256 * 8), {"FLACGPU_BATCH_FRAMES": "4"})        # last block exactly full, batch exactly full
    check(dict(BASE, samples=1), {"FLACGPU_BATCH_FRAMES": "4"})
    check(dict(BASE, samples=256), {"FLACGPU_BATCH_FRAMES": "1"})


def test_out_of_range_sample_is_refused_by_every_thread():
    """a value outside the stream's range fails the call (stream_encoder.c:2544-2547) wherever in the call it sits"""
    _build()
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, ctypes as C
import flac_api
lib = flac_api.lib_for("gpu")
bad = 0
for pos in (0, 100000, 262143, 300000, 524287):
    e = lib.FLAC__stream_encoder_new()
    lib.FLAC__stream_encoder_set_channels(e, 2); lib.FLAC__stream_encoder_set_bits_per_sample(e, 16); lib.FLAC__stream_encoder_set_sample_rate(e, 44100)
    sink = flac_api.Sink()
    assert lib.FLAC__stream_encoder_init_stream(e, *sink.callbacks(), None) == 0
    pcm = np.zeros((262144, 2), dtype=np.int32)
    pcm.reshape(-1)[pos] = 40000
    ok = lib.FLAC__stream_encoder_process_interleaved(e, pcm.ctypes.data, len(pcm))
    st = lib.FLAC__stream_encoder_get_state(e)
    bad += (not ok) and st == 5                                   # FLAC__STREAM_ENCODER_CLIENT_ERROR
    lib.FLAC__stream_encoder_finish(e); lib.FLAC__stream_encoder_delete(e)
print(bad)
''' % {"root": ROOT}
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["FLACGPU_STAGE_THREADS"] = "4"
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert out.stdout.strip() == b"5"


# ---- bring-up on its own thread (HIP runtime + engine + page-locking while the caller already stages) -------------------
def test_slow_bring_up_changes_nothing():
    """the engine appears 0.3 s after init returned: the caller has staged several batches by then, the MD5 chain ran ahead"""
    N = 4096
    case = dict(BASE, blocksize=N, samples=N * 100 + 17, chunk=N * 10, seed=8)
    check(case, {"FLACGPU_BATCH_FRAMES": "16", "FAKE_ENGINE_CREATE_DELAY_US": "300000"})
    check(case, {"FLACGPU_BATCH_FRAMES": "16", "FAKE_ENGINE_CREATE_DELAY_US": "100000", "FLACGPU_SYNC_INIT": "1"})
    check(case, {"FLACGPU_BATCH_FRAMES": "16", "FAKE_ENGINE_FAIL_REGISTER": "1"})          # slots that cannot be page-locked still work


FAIL_CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import flac_api
lib = flac_api.lib_for("gpu")
e = lib.FLAC__stream_encoder_new()
lib.FLAC__stream_encoder_set_channels(e, 2); lib.FLAC__stream_encoder_set_bits_per_sample(e, 16); lib.FLAC__stream_encoder_set_sample_rate(e, 44100)
sink = flac_api.Sink()
st = lib.FLAC__stream_encoder_init_stream(e, *sink.callbacks(), None)
state_after_init = lib.FLAC__stream_encoder_get_state(e)
ok_process = ok_finish = None
if st == 0 and sys.argv[1] == "encode":
    pcm = np.zeros((4096 * 40, 2), dtype=np.int32)
    ok_process = bool(lib.FLAC__stream_encoder_process_interleaved(e, pcm.ctypes.data, len(pcm)))
    state_after_process = lib.FLAC__stream_encoder_get_state(e)
    ok_finish = bool(lib.FLAC__stream_encoder_finish(e))
    print(st, state_after_init, ok_process, state_after_process, ok_finish, len(sink.buf.getvalue()))
elif st == 0 and sys.argv[1] == "encode_records":
    # many small calls, so that batches are submitted while earlier ones are in flight; then look at the records that came out
    import struct
    pcm = np.zeros((4096 * 40, 2), dtype=np.int32)
    ok_process = True
    for a in range(0, len(pcm), 4096):
        part = np.ascontiguousarray(pcm[a:a + 4096])
        if not lib.FLAC__stream_encoder_process_interleaved(e, part.ctypes.data, len(part)):
            ok_process = False
            break
    state_after_process = lib.FLAC__stream_encoder_get_state(e)
    ok_finish = bool(lib.FLAC__stream_encoder_finish(e))
    data = sink.buf.getvalue()
    body = data[data.find(b"FK\0\0"):] if b"FK\0\0" in data else b""
    recs, pos, garbage = [], 0, 0
    while pos < len(body):
        if body[pos:pos + 4] != b"FK\0\0" or pos + 16 > len(body):
            garbage = 1
            break
        fn, n, h = struct.unpack("<III", body[pos + 4:pos + 16])
        recs.append(fn)
        pos += 16 + fn %% 5
    garbage |= int(any(sz > (1 << 20) for sz, _, _ in sink.calls))
    print(st, state_after_init, ok_process, state_after_process, ok_finish, len(data), len(recs), recs == list(range(len(recs))), garbage)
else:
    print(st, state_after_init)
lib.FLAC__stream_encoder_delete(e)            # with the bring-up thread possibly still in flight
'''


def _fail_child(mode, env_extra):
    _build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", FAIL_CHILD % {"root": ROOT}, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return out.stdout.decode().split(), out.stderr.decode()


def test_engine_failure_is_loud_wherever_it_surfaces():
    # no device node, or synchronous bring-up asked for: init itself fails (ENCODER_ERROR = 1) with an error state
    for env in ({"FAKE_ENGINE_FAIL_CREATE": "1", "FAKE_ENGINE_NO_PROBE": "1"}, {"FAKE_ENGINE_FAIL_CREATE": "1", "FLACGPU_SYNC_INIT": "1"}):
        out, err = _fail_child("encode", env)
        assert out[0] == "1" and out[1] != "0" and "cannot create the GPU frame engine" in err
    # a device node that opens but an engine that does not come up: init has returned OK by then; the first call that needs a
    # frame fails, the state is an error state, stderr says why -- and NOT ONE BYTE reached the client's output: the stream's head
    # ("fLaC", STREAMINFO, ...) is held back until the engine is known to be there (no half-written file)
    out, err = _fail_child("encode", {"FAKE_ENGINE_FAIL_CREATE": "1", "FLACGPU_BATCH_FRAMES": "8", "FAKE_ENGINE_CREATE_DELAY_US": "50000"})
    st, s_init, ok_process, s_proc, ok_finish, written = out
    assert st == "0" and s_init == "0"
    assert ok_process == "False" and s_proc != "0" and "cannot create the GPU frame engine" in err
    assert ok_finish == "True"               # the reference's rule: finish() after a failed process() has nothing left to fail (:1649)
    assert written == "0"
    # a stream shorter than one batch meets the failure in finish()
    out, err = _fail_child("encode", {"FAKE_ENGINE_FAIL_CREATE": "1", "FLACGPU_BATCH_FRAMES": "64"})
    st, s_init, ok_process, s_proc, ok_finish, written = out
    assert (st, s_init, ok_process, s_proc, ok_finish, written) == ("0", "0", "True", "0", "False", "0") and "cannot create the GPU frame engine" in err
    # the synchronous flavours fail in init, before any byte as well
    out, err = _fail_child("encode", {"FAKE_ENGINE_FAIL_CREATE": "1", "FLACGPU_SYNC_INIT": "1"})
    assert out[0] == "1"


def test_empty_stream_with_asynchronous_bring_up():
    """no sample at all: the head of the stream goes out from finish(), once the engine is up"""
    case = dict(BASE, samples=0)
    data = run_case(case, {"FAKE_ENGINE_CREATE_DELAY_US": "100000"})
    assert data[:4] == b"fLaC" and data[4] == 0 and data[42] == 0x84
    assert data == run_case(case, {"FLACGPU_SYNC_INIT": "1"})


OGG_CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import flac_api
pcm = np.zeros((int(sys.argv[1]), 2), dtype=np.int32)
data, sink = flac_api.encode("gpu", pcm, 16, 44100, level=5, settings=[("set_blocksize", 256), ("set_ogg_serial_number", 77)], ogg=True)
sys.stdout.buffer.write(data)
'''


@pytest.mark.parametrize("samples", [100, 256, 300])
def test_ogg_stream_of_one_frame_with_asynchronous_bring_up(samples):
    """the stream's head is written from inside the first frame's emit when the engine came up beside init_*(); for a
    stream whose first frame is also its last, the metadata packets must not inherit that frame's is_last_block (the BOS page
    carried the EOS flag).  Same bytes as the synchronous bring-up; flags: BOS on the first page only, EOS on the last only."""
    _build()
    outs = []
    for extra in ({"FAKE_ENGINE_CREATE_DELAY_US": "50000"}, {"FLACGPU_SYNC_INIT": "1"}):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", OGG_CHILD % {"root": ROOT}, str(samples)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        outs.append(out.stdout)
    assert outs[0] == outs[1]
    flags, pos, d = [], 0, outs[0]
    while pos < len(d):
        assert d[pos:pos + 4] == b"OggS"
        flags.append(d[pos + 5])
        nseg = d[pos + 26]
        pos += 27 + nseg + sum(d[pos + 27:pos + 27 + nseg])
    assert flags[0] == 2 and all(f == 0 for f in flags[1:-1]) and flags[-1] == 4


def test_delete_while_the_engine_is_still_coming_up():
    out, err = _fail_child("nothing", {"FAKE_ENGINE_CREATE_DELAY_US": "200000"})
    assert out == ["0", "0"]


# ---- the parked engine: one stream after another in one process ------------------------------------------------------------
PARK_CHILD = r'''
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import flac_api
fake = C.CDLL(os.path.join(%(fake)r, "libflacgpu.so"))
jobs = json.loads(sys.argv[1])
out = []
for job in jobs:
    rng = np.random.default_rng(job["seed"])
    pcm = rng.integers(-32768, 32768, size=(job["samples"], 2), dtype=np.int64).astype(np.int32)
    settings = [("set_blocksize", job["blocksize"])] + [tuple(x) for x in job.get("settings", [])]
    data, sink = flac_api.encode("gpu", pcm, 16, 44100, level=job.get("level", 5), chunk=5000, settings=settings,
                                 total_samples_estimate=job.get("estimate"))
    out.append({"len": len(data), "md5": data[26:42].hex(), "creates": fake.fake_engine_creates(), "destroys": fake.fake_engine_destroys(),
                "frames": data.count(bytes([70, 75, 0, 0]))})
print(json.dumps(out))
'''


def _park(jobs, env_extra=()):
    import json
    _build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["FLACGPU_BATCH_FRAMES"] = "16"
    env.update(dict(env_extra))
    out = subprocess.run([sys.executable, "-c", PARK_CHILD % {"root": ROOT, "fake": FAKE_DIR}, json.dumps(jobs)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout)


def _md5_of(job):
    rng = np.random.default_rng(job["seed"])
    pcm = rng.integers(-32768, 32768, size=(job["samples"], 2), dtype=np.int64).astype(np.int32)
    return hashlib.md5(pcm.astype("<i2").tobytes()).hexdigest()


def test_next_stream_with_the_same_settings_gets_the_parked_engine():
    a = dict(seed=1, samples=256 * 40 + 3, blocksize=256)
    b = dict(seed=2, samples=256 * 70, blocksize=256)
    c = dict(seed=3, samples=256 * 5, blocksize=256, estimate=256 * 5)        # wants 5-frame batches: the parked 16-frame engine serves it
    r = _park([a, b, c])
    assert [x["creates"] for x in r] == [1, 1, 1] and r[-1]["destroys"] == 0
    assert [x["md5"] for x in r] == [_md5_of(a), _md5_of(b), _md5_of(c)]
    assert [x["frames"] for x in r] == [41, 70, 5]


def test_other_settings_get_their_own_engine_and_replace_the_parked_one():
    a = dict(seed=1, samples=256 * 20, blocksize=256)
    b = dict(seed=2, samples=512 * 20, blocksize=512)                         # other block size
    c = dict(seed=3, samples=512 * 20, blocksize=512, level=8)                # other preset: other windows, orders
    r = _park([a, b, c, c])
    assert [x["creates"] for x in r] == [1, 2, 3, 3]
    assert [x["destroys"] for x in r] == [0, 1, 2, 2]                        # the older engine goes when a newer one is parked
    assert [x["md5"] for x in r] == [_md5_of(a), _md5_of(b), _md5_of(c), _md5_of(c)]
    # a bigger stream than the parked engine was built for: a new engine
    small = dict(seed=5, samples=256 * 4, blocksize=256, estimate=256 * 4)
    r = _park([small, a])
    assert [x["creates"] for x in r] == [1, 2]


def test_parking_can_be_switched_off():
    a = dict(seed=1, samples=256 * 20, blocksize=256)
    r = _park([a, a], {"FLACGPU_ENGINE_CACHE": "0"})
    assert [x["creates"] for x in r] == [1, 2] and [x["destroys"] for x in r] == [1, 2]


DROPIN_FLAC = os.path.join(ROOT, "oracle", "_ref", "dropin", "flac")


@pytest.mark.skipif(not os.path.exists(DROPIN_FLAC), reason="oracle/_ref/dropin/flac not built (no /root/reference on this box)")
def test_client_that_frees_its_metadata_right_after_init(tmp_path):
    """the reference's `flac` tool deletes its metadata objects as soon as init_*() has returned (src/flac/encode.c:2172) -- the
    head of the stream is serialised inside init, even when it is written later (asynchronous bring-up): same file either way"""
    import wave
    _build()
    rng = np.random.default_rng(3)
    pcm = rng.integers(-20000, 20000, size=(4096 * 9 + 500, 2), dtype=np.int64).astype("<i2")
    wav = str(tmp_path / "a.wav")
    w = wave.open(wav, "wb"); w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes()); w.close()
    outs = []
    for extra in ({"FAKE_ENGINE_CREATE_DELAY_US": "100000"}, {"FLACGPU_SYNC_INIT": "1"}):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = FAKE_DIR + os.pathsep + os.path.join(ROOT, "flac_amd", "lib") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
        env.update(extra)
        out = str(tmp_path / ("o%d.flac" % len(outs)))
        r = subprocess.run([DROPIN_FLAC, "-s", "-f", "-5", "-T", "TITLE=x", "-S", "4x", "--padding=100", "-o", out, wav], env=env, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0][:4] == b"fLaC"


# ---- the ring of batch slots on the engine's asynchronous entry (round 3) ----------------------------------------------------------
@pytest.mark.parametrize("md5", [0, 1])
@pytest.mark.parametrize("batch,delay_us", [(1, 0), (2, 3000), (3, 500), (5, 20000)])
def test_ring_of_batches_in_flight(batch, delay_us, md5):
    """a slow engine (the caller runs NSLOT - 1 batches ahead and then waits), a fast one, tiny batches that lap the ring many times:
    same records in the same order, the MD5 chain still over exactly the stream's bytes.  The fake engine reads a batch's raw bytes
    only when it is collected: a slot reused too early would show in the records' checksums."""
    case = dict(BASE, md5=md5, samples=256 * 41 + 13, chunk=700)
    env = {"FLACGPU_BATCH_FRAMES": str(batch)}
    if delay_us:
        env["FAKE_ENGINE_DELAY_US"] = str(delay_us)
    check(case, env)


@pytest.mark.parametrize("nth", [1, 2, 3, 4])
def test_one_refused_submission_among_good_ones_fails_the_stream_cleanly(nth):
    """only the n-th submission is refused while earlier batches are in flight and later ones would go through.  The
    frames in front of the failed batch are delivered in order (whole records, none from a slot that was never filled), process()
    fails, finish() returns (no hang in the drain loop)."""
    out, err = _fail_child("encode_records", {"FAKE_ENGINE_FAIL_SUBMIT_NTH": str(nth), "FLACGPU_BATCH_FRAMES": "4", "FLACGPU_SYNC_INIT": "1",
                                               "FAKE_ENGINE_DELAY_US": "2000"})
    st, s_init, ok_process, s_proc, ok_finish, written, nrec, in_order, garbage = out
    assert st == "0" and ok_process == "False" and s_proc != "0" and "the GPU frame engine failed" in err
    assert garbage == "0" and in_order == "True"
    assert int(nrec) == 4 * (nth - 1)                  # exactly the batches in front of the refused one


def test_a_refused_submission_fails_the_stream():
    out, err = _fail_child("encode", {"FAKE_ENGINE_FAIL_SUBMIT": "1", "FLACGPU_BATCH_FRAMES": "8", "FLACGPU_SYNC_INIT": "1"})
    st, s_init, ok_process, s_proc, ok_finish, written = out
    assert st == "0" and ok_process == "False" and s_proc != "0" and "the GPU frame engine failed" in err
