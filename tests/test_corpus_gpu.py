"""BASELINE.json config 5 at test scale: the corpus job (flac_amd/corpus.py: shard_range -> raw staging on the device ->
batched encode -> ordered gather over the `nccl` backend -> one .flac with STREAMINFO/MD5) against the unmodified
reference encoding the same samples through its file API.  One GPU here, so the process group has one rank
(--force-dist): the RCCL code path, the batch splitting with appended offsets and the short last block are what is
exercised; the 8-rank geometry is covered on CPU by tests/test_dist_cpu.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tmp_path, name, *args):
    out = str(tmp_path / (name + ".flac"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "flac_amd.corpus", "--out", out] + list(args), cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    with open(out, "rb") as f:
        return line, f.read()


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/libFLAC_ref.so not built on this box")
def test_corpus_job_equals_the_reference_file(tmp_path):
    from flac_amd import corpus as co
    samples = 180 * 44100 + 1234                      # three minutes and a short last block: 1939 frames
    base = co.base_clip()
    nfr = (samples + co.BLOCK - 1) // co.BLOCK
    pcm = co.host_frames(base, 0, nfr)[:samples].astype(np.int32)
    want = po.ref_encode_file(pcm, 16, 44100, 8, str(tmp_path / "ref.flac"), do_md5=1)

    single, a = _run(tmp_path, "single", "--samples", str(samples))
    assert single["crc16_bad_frame"] == -1 and single["frames"] == nfr
    assert a == want, "single-rank corpus file differs from the reference's"
    # the process-group path (nccl backend, one rank), batches of 700 frames: three appended batches, the last one short
    d, b = _run(tmp_path, "dist", "--samples", str(samples), "--force-dist", "--batch-frames", "700")
    assert b == want, "gathered corpus file differs from the reference's"
    assert d["crc16_bad_frame"] == -1
    # the shared-pinned-host-buffer gather
    h, c = _run(tmp_path, "host", "--samples", str(samples), "--force-dist", "--gather", "host", "--batch-frames", "1024")
    assert c == want


@pytest.mark.parametrize("md5", ["device", "host", "from-host"])
@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref/libFLAC_ref.so not built on this box")
def test_corpus_as_tracks_equals_the_reference_files(tmp_path, md5):
    """--tracks: the corpus as separate streams, frame numbers restarting per track, a STREAMINFO and an MD5 per track (the digests
    on the device, one lane per track over the staged sample bytes, or on the host, eight chains at a time): every track file is
    the reference's file for that track's samples"""
    from flac_amd import corpus as co
    samples = 60 * 44100 + 777
    ntracks = 11                                       # 646 frames in 11 tracks: 59 / 58 frames each, the last one with the short block
    base = co.base_clip()
    nfr = (samples + co.BLOCK - 1) // co.BLOCK
    pcm = co.host_frames(base, 0, nfr)[:samples].astype(np.int32)
    out = str(tmp_path / "shelf")
    env = dict(os.environ)
    # from-host: the job's input in page-locked host memory, copied to the device track by track beside the encodes and hashed by host
    # threads straight from that buffer (the default since round 5); device / host: the input generated in HBM (round 4's job)
    how = ["--input", "host", "--md5-threads", "3"] if md5 == "from-host" else ["--input", "device", "--md5-threads", "2", "--md5", md5]
    r = subprocess.run([sys.executable, "-m", "flac_amd.corpus", "--out", out, "--samples", str(samples), "--tracks", str(ntracks)] + how,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["tracks"] == ntracks and line["tracks_with_a_bad_crc16"] == 0
    import hashlib
    for t, (lo, hi) in enumerate(co.track_ranges(nfr, ntracks)):
        assert line["track_md5"][t] == hashlib.md5(pcm[lo * co.BLOCK:min(hi * co.BLOCK, samples)].astype("<i2").tobytes()).hexdigest(), t
    for t, (lo, hi) in enumerate(co.track_ranges(nfr, ntracks)):
        want = po.ref_encode_file(pcm[lo * co.BLOCK:min(hi * co.BLOCK, samples)], 16, 44100, 8, str(tmp_path / "ref.flac"), do_md5=1)
        with open("%s.%04d.flac" % (out, t), "rb") as f:
            assert f.read() == want, "track %d" % t
