"""flac_amd.corpus -- batch encode of a long corpus into ONE .flac, sharded over the GPUs of a node
(BASELINE.json config 5: `flac -8` on 10 h of 44.1 kHz/16-bit stereo, RCCL gather).

Every rank takes the contiguous frame range shard_range() gives it, with the frame numbers fixed up front
(frames are independent: stream_encoder.c:3778-3807 made even the loose mid/side decision per frame), stages its
shard from raw 16-bit file bytes on the device (flacgpu_stage_raw_device), encodes it in batches
(flacgpu_encode_batch_device) into one device buffer, and the job ends with the ordered gather of the bitstream:
    --gather rccl   flac_amd.dist.ordered_gather: point-to-point RCCL transfers to rank 0's HBM (the north star's form)
    --gather host   flac_amd.dist.HostShmGather: every rank copies its shard over its own PCIe link into one shared
                    pinned host buffer at the scanned offset
Rank 0 then writes the stream the way FLAC__stream_encoder_finish leaves a file (stream_encoder.c:3139-3246):
"fLaC", STREAMINFO with min/max frame size, total samples and the MD5 of the interleaved input (computed by a host
thread while the GPUs work: one serial chain, md5.c:497), the VORBIS_COMMENT block with the vendor string, the frames.

    python -m flac_amd.corpus --hours 10 --out /tmp/corpus.flac
    python -m flac_amd.corpus --gpus 8 --hours 10 ...          (starts its eight ranks itself: flac_amd.dist.ensure_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m flac_amd.corpus --hours 10 ...

Prints one JSON line (rank 0).  The synthetic corpus is a 512-frame music-like clip repeated with an integer gain /
offset per repetition (integer arithmetic only, so that the device and the host produce the same samples).
"""
import argparse
import hashlib
import json
import os
import struct
import sys
import threading
import time

import numpy as np

RATE, BPS, CH, BLOCK = 44100, 16, 2, 4096
BASE_FRAMES = 512
VENDOR = b"reference libFLAC 1.5.0 20250211"        # format.c:57 (the host API writes the same string, host/stream_encoder.c)


def base_clip(seed=1234):
    """the 48 s clip the corpus is made of: int16 [BASE_FRAMES*BLOCK, 2] (flac_amd/signals.py: tones + coloured noise)"""
    from . import signals
    return np.ascontiguousarray(signals.music(BASE_FRAMES * BLOCK, CH, BPS, seed=seed, rate=RATE).astype(np.int16))


def rep_params(rep):
    """gain numerator (/256) and offset of repetition `rep` of the clip"""
    return 256 - 18 * (rep % 8), (rep % 5) - 2


def host_frames(base, f0, f1):
    """frames [f0, f1) of the corpus as int16 [(f1-f0)*BLOCK, 2] (numpy, integer arithmetic)"""
    out = np.empty(((f1 - f0) * BLOCK, CH), dtype=np.int16)
    f = f0
    while f < f1:
        rep, b0 = divmod(f, BASE_FRAMES)
        nb = min(BASE_FRAMES - b0, f1 - f)
        g, o = rep_params(rep)
        x = base[b0 * BLOCK:(b0 + nb) * BLOCK].astype(np.int32)
        y = np.clip(((x * g) >> 8) + o, -32768, 32767).astype(np.int16)
        out[(f - f0) * BLOCK:(f - f0 + nb) * BLOCK] = y
        f += nb
    return out


def device_frames(base_dev, f0, f1, out):
    """the same on the device (torch int32 arithmetic), into out: int16 [(f1-f0)*BLOCK, 2]"""
    import torch
    f = f0
    while f < f1:
        rep, b0 = divmod(f, BASE_FRAMES)
        nb = min(BASE_FRAMES - b0, f1 - f)
        g, o = rep_params(rep)
        x = base_dev[b0 * BLOCK:(b0 + nb) * BLOCK].to(torch.int32)
        y = torch.clamp(torch.bitwise_right_shift(x * g, 8) + o, -32768, 32767).to(torch.int16)
        out[(f - f0) * BLOCK:(f - f0 + nb) * BLOCK] = y
        f += nb


def stream_header(total_samples, min_frame, max_frame, md5):
    """"fLaC" + STREAMINFO + the default VORBIS_COMMENT, as init_stream writes and finish patches them
    (stream_encoder.c:1335-1425, 3139-3246; stream_encoder_framing.c:46-243)"""
    si = struct.pack(">HH", BLOCK, BLOCK) + min_frame.to_bytes(3, "big") + max_frame.to_bytes(3, "big")
    si += ((RATE << 44) | ((CH - 1) << 41) | ((BPS - 1) << 36) | (total_samples if total_samples < (1 << 36) else 0)).to_bytes(8, "big")
    si += md5
    assert len(si) == 34
    vc = struct.pack("<I", len(VENDOR)) + VENDOR + struct.pack("<I", 0)
    return b"fLaC" + bytes([0]) + len(si).to_bytes(3, "big") + si + bytes([0x80 | 4]) + len(vc).to_bytes(3, "big") + vc


def encode_shard(eng, base_dev, lo, hi, batch_frames, dev, out, fb_all, enc_stream, last_frame=-1, tail=0):
    """Stage + encode frames [lo, hi) in batches.  out: uint8 device buffer for the whole shard, fb_all: int32 [hi-lo].
    Frame `last_frame` (the last of the stream) has only `tail` samples when tail != 0.
    Returns the shard's byte total (host int; one host read per batch: the next batch's offset)."""
    import torch
    import flac_amd
    fmt = flac_amd.raw_format(16)
    raw = torch.empty((batch_frames * BLOCK, CH), dtype=torch.int16, device=dev)
    pcm = torch.empty((batch_frames * BLOCK, CH), dtype=torch.int32, device=dev)
    total_t = torch.zeros(1, dtype=torch.int64, device=dev)
    off = 0
    f = lo
    with torch.cuda.stream(enc_stream):
        while f < hi:
            nf = min(batch_frames, hi - f)
            device_frames(base_dev, f, f + nf, raw)
            short = tail if (tail and f + nf - 1 == last_frame) else 0
            eng.stage_raw_device(raw.data_ptr(), fmt, nf * BLOCK - (BLOCK - short if short else 0), pcm.data_ptr(), None, enc_stream.cuda_stream)
            eng.encode_device(pcm.data_ptr(), nf, out.data_ptr() + off, out.numel() - off, fb_all.data_ptr() + 4 * (f - lo), total_t.data_ptr(),
                              first_frame_number=f, tail=short, stream=enc_stream.cuda_stream)
            off += int(total_t.item())
            f += nf
    return off


def md5_many(buffers):
    """MD5 of each buffer (bytes-like, contiguous), eight chains at a time on one thread (flac_amd/csrc/host/md5.c: AVX2, a chain per
    32-bit lane): a corpus of many tracks needs one digest per track, and one chain alone is serial"""
    import ctypes as C
    import flac_amd
    lib = flac_amd.engine.load_host()
    lib.flacgpu_host_md5_many.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_void_p]
    lib.flacgpu_host_md5_many.restype = None
    n = len(buffers)
    views = [np.frombuffer(b, dtype=np.uint8) for b in buffers]
    ptrs = (C.c_void_p * n)(*[v.ctypes.data if v.size else 0 for v in views])
    lens = (C.c_size_t * n)(*[v.size for v in views])
    out = np.zeros((n, 16), dtype=np.uint8)
    lib.flacgpu_host_md5_many(ptrs, lens, n, out.ctypes.data)
    return [out[i].tobytes() for i in range(n)]


def md5_many_mt(ptrs, lens, nthreads):
    """MD5 of len(ptrs) byte ranges of host memory on `nthreads` host threads (flac_amd/csrc/host/md5.c: flacgpu_host_md5_many_mt --
    groups of sixteen chains per AVX-512 register, or eight per AVX2 register, a group per work item); the GIL is released inside"""
    import ctypes as C
    import flac_amd
    lib = flac_amd.engine.load_host()
    lib.flacgpu_host_md5_many_mt.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_void_p, C.c_uint32]
    lib.flacgpu_host_md5_many_mt.restype = None
    n = len(ptrs)
    cp = (C.c_void_p * n)(*ptrs)
    cl = (C.c_size_t * n)(*lens)
    out = np.zeros((n, 16), dtype=np.uint8)
    lib.flacgpu_host_md5_many_mt(cp, cl, n, out.ctypes.data, max(1, int(nthreads)))
    return [out[i].tobytes() for i in range(n)]


def usable_cpus():
    """CPUs this process may really use (affinity and cgroup quota: the GPU boxes show 256 hardware threads to a container allowed 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def host_corpus(base, F, total_samples):
    """the whole corpus as the job finds it when its input comes from the host -- the tracks' sample bytes as read from their files --
    in ONE page-locked buffer: int16 [F*BLOCK, 2] (the frames behind total_samples are never read)"""
    import torch
    buf = torch.empty((F * BLOCK, CH), dtype=torch.int16, pin_memory=True)
    arr = buf.numpy()
    cache = {}
    for f in range(0, F, BASE_FRAMES):
        nb = min(BASE_FRAMES, F - f)
        key = rep_params(f // BASE_FRAMES)
        if key not in cache:
            cache[key] = host_frames(base, f, f + BASE_FRAMES)
        arr[f * BLOCK:(f + nb) * BLOCK] = cache[key][:nb * BLOCK]
    return buf


def check_hbm(dev, need_bytes, what):
    """a clear refusal instead of an out-of-memory error half way (the corpus-sized buffers were allocated unchecked)"""
    import torch
    free, total = torch.cuda.mem_get_info(dev)
    if need_bytes + (1 << 30) > free:
        raise SystemExit("flac_amd.corpus: %s needs %.1f GB of HBM, %.1f GB free of %.1f -- encode the corpus in parts (--hours) or from the host (--input host)"
                         % (what, need_bytes / 1e9, free / 1e9, total / 1e9))


def tracks_device_buffers(eng, F, ntracks, dev):
    """the device side of encode_tracks_from_host -- the corpus's sample bytes, the worst-case frames, one track's int32 block, frame
    lengths, byte totals: 13 GB of the 288 for ten hours -- allocated once, before the job's clock starts (hipMalloc of that much takes
    0.1-0.4 s on a fresh process; a service that encodes corpora keeps them)"""
    import torch
    ranges = track_ranges(F, ntracks)
    maxf = max(hi - lo for lo, hi in ranges)
    slot = eng.max_output_bytes(1)
    check_hbm(dev, F * BLOCK * CH * 2 + F * slot + maxf * BLOCK * CH * 4, "the corpus (sample bytes + worst-case frames)")
    return {"raw_all": torch.empty((F * BLOCK, CH), dtype=torch.int16, device=dev), "pcm": torch.empty((maxf * BLOCK, CH), dtype=torch.int32, device=dev),
            "out": torch.empty(F * slot, dtype=torch.uint8, device=dev), "fb_all": torch.zeros(F, dtype=torch.int32, device=dev),
            "totals": torch.zeros(ntracks, dtype=torch.int64, device=dev)}


def encode_tracks_from_host(eng, hbuf, F, total_samples, ntracks, dev, enc_stream, want_md5=True, md5_threads=0, bufs=None):
    """The corpus as `ntracks` separate streams, input in page-locked host memory (the tracks' sample bytes as their files hold them).
    The reference hashes what it encodes on the way in (FLAC__MD5Accumulate per block, stream_encoder.c:3666-3686 -> md5.c:497); here
    the two run side by side: host threads hash the tracks straight from the input buffer (no copy, no "prepare": sixteen or eight
    chains per thread, flacgpu_host_md5_many_mt) while the copy engine moves them to the device track by track and the engine stages
    and encodes each as its bytes arrive.  One read of the tracks' byte totals at the end."""
    import torch
    import flac_amd
    fmt = flac_amd.raw_format(16)
    ranges = track_ranges(F, ntracks)
    tail = total_samples - (F - 1) * BLOCK
    tail = 0 if tail == BLOCK else tail
    slot = eng.max_output_bytes(1)
    bufs = bufs or tracks_device_buffers(eng, F, ntracks, dev)
    raw_all, pcm, out, fb_all, totals = bufs["raw_all"], bufs["pcm"], bufs["out"], bufs["fb_all"], bufs["totals"]
    starts = [lo * slot for lo, _ in ranges]
    copy_stream = torch.cuda.Stream()
    arrived = [torch.cuda.Event() for _ in ranges]
    harr = hbuf.numpy()
    t0 = time.perf_counter()
    box = {}
    th = None
    if want_md5:
        ptrs = [harr[lo * BLOCK:].ctypes.data if hi > lo else 0 for lo, hi in ranges]
        lens = [max(0, (min(hi * BLOCK, total_samples) - lo * BLOCK)) * CH * 2 if hi > lo else 0 for lo, hi in ranges]

        def hash_all():
            tm = time.perf_counter()
            box["digests"] = md5_many_mt(ptrs, lens, md5_threads or usable_cpus())
            box["seconds"] = time.perf_counter() - tm
        th = threading.Thread(target=hash_all)
        th.start()
    with torch.cuda.stream(copy_stream):
        for t, (lo, hi) in enumerate(ranges):
            if hi > lo:
                raw_all[lo * BLOCK:hi * BLOCK].copy_(hbuf[lo * BLOCK:hi * BLOCK], non_blocking=True)
            arrived[t].record(copy_stream)
    with torch.cuda.stream(enc_stream):
        for t, (lo, hi) in enumerate(ranges):
            nf = hi - lo
            if nf == 0:
                continue
            enc_stream.wait_event(arrived[t])
            short = tail if (tail and hi == F) else 0
            eng.stage_raw_device(raw_all[lo * BLOCK:].data_ptr(), fmt, nf * BLOCK - (BLOCK - short if short else 0), pcm.data_ptr(), None, enc_stream.cuda_stream)
            eng.encode_device(pcm.data_ptr(), nf, out.data_ptr() + starts[t], nf * slot, fb_all.data_ptr() + 4 * lo, totals.data_ptr() + 8 * t,
                              first_frame_number=0, tail=short, stream=enc_stream.cuda_stream)
    copy_stream.synchronize()
    t_copy = time.perf_counter() - t0
    enc_stream.synchronize()
    t_enc = time.perf_counter() - t0
    if th:
        th.join()
    t_all = time.perf_counter() - t0
    tot_h = totals.cpu().tolist()
    digests = box.get("digests", [bytes(16)] * ntracks)
    fbs = fb_all.cpu().numpy().astype(np.uint32)
    streams = []
    for t, (lo, hi) in enumerate(ranges):
        nsamp = min(hi * BLOCK, total_samples) - lo * BLOCK if hi > lo else 0
        f = fbs[lo:hi]
        header = stream_header(nsamp, int(f.min()) if f.size else 0, int(f.max()) if f.size else 0, digests[t])
        streams.append((header, out[starts[t]:starts[t] + int(tot_h[t])], f))
    return streams, {"encode_seconds": t_enc, "h2d_seconds": t_copy, "md5_prepare_seconds": 0.0, "md5_seconds": box.get("seconds", 0.0), "hash_and_encode_seconds": t_all,
                     "host_reads_of_totals": 1, "md5_threads": (md5_threads or usable_cpus()) if want_md5 else 0}


def track_ranges(F, ntracks):
    """frames [lo, hi) of each track: equal shares of the corpus, earlier tracks take the remainder"""
    base, rem = divmod(F, ntracks)
    out, lo = [], 0
    for t in range(ntracks):
        hi = lo + base + (1 if t < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def encode_tracks(eng, base, base_dev, F, total_samples, ntracks, dev, enc_stream, want_md5=True, md5_threads=1, md5_where="host"):
    """The corpus as `ntracks` separate streams (BASELINE config 5 as an album shelf rather than one file): every track's frames are
    numbered from 0 and get their own STREAMINFO (total samples, min/max frame size, MD5 of the track's samples).  One engine call
    per track (frame numbers of a call are consecutive).  Returns (list of (header, frames uint8 tensor view), timings)."""
    import torch
    import flac_amd
    fmt = flac_amd.raw_format(16)
    ranges = track_ranges(F, ntracks)
    tail = total_samples - (F - 1) * BLOCK
    tail = 0 if tail == BLOCK else tail
    maxf = max(hi - lo for lo, hi in ranges)
    on_device = want_md5 and md5_where == "device"
    # md5_where == "device": the whole corpus's sample bytes stay staged in HBM (6.4 GB for ten hours) and are hashed there, one lane
    # per track (flacgpu_md5.hip), BEHIND the encodes on their stream -- no host copy of the samples, no host threads
    check_hbm(dev, (F * BLOCK * CH * 2 if on_device else maxf * BLOCK * CH * 2) + F * eng.max_output_bytes(1) + maxf * BLOCK * CH * 4, "the corpus (sample bytes + worst-case frames)")
    raw_all = torch.empty((F * BLOCK, CH), dtype=torch.int16, device=dev) if on_device else None
    raw = None if on_device else torch.empty((maxf * BLOCK, CH), dtype=torch.int16, device=dev)
    pcm = torch.empty((maxf * BLOCK, CH), dtype=torch.int32, device=dev)
    # Every track is a stream of its own, so nothing has to be packed behind its predecessor: track t is encoded to where its frames
    # would start if every frame in front took its worst case (a few GB of the 288 for a ten-hour corpus), the engine calls go out
    # back to back, and the tracks' byte totals are read ONCE, at the end (round 3 read one per track before it could place the next:
    # encode_seconds was host round trips, not the engine -- ADVICE r03)
    slot = eng.max_output_bytes(1)
    out = torch.empty(F * slot, dtype=torch.uint8, device=dev)
    fb_all = torch.zeros(F, dtype=torch.int32, device=dev)
    totals = torch.zeros(ntracks, dtype=torch.int64, device=dev)
    starts = [lo * slot for lo, _ in ranges]
    t0 = time.perf_counter()
    with torch.cuda.stream(enc_stream):
        if on_device:
            # the corpus's sample bytes, all of them, before the first encode
            device_frames(base_dev, 0, F, raw_all)
        for t, (lo, hi) in enumerate(ranges):
            nf = hi - lo
            if nf == 0:
                continue
            traw = raw_all[lo * BLOCK:hi * BLOCK] if on_device else raw
            if not on_device:
                device_frames(base_dev, lo, hi, traw)
            short = tail if (tail and hi == F) else 0
            eng.stage_raw_device(traw.data_ptr(), fmt, nf * BLOCK - (BLOCK - short if short else 0), pcm.data_ptr(), None, enc_stream.cuda_stream)
            eng.encode_device(pcm.data_ptr(), nf, out.data_ptr() + starts[t], nf * slot, fb_all.data_ptr() + 4 * lo, totals.data_ptr() + 8 * t,
                              first_frame_number=0, tail=short, stream=enc_stream.cuda_stream)
    t_dev_md5 = 0.0
    dev_digests = None
    if on_device:
        offs_b = [lo * BLOCK * CH * 2 for lo, _ in ranges]
        lens_b = [max(0, (min(hi * BLOCK, total_samples) - lo * BLOCK)) * CH * 2 if hi > lo else 0 for lo, hi in ranges]
        # behind the encodes, on their stream (beside them, on a stream of its own, the two wavefronts of 120 chains took twice as long
        # and the job with them: profiles/archive/r04_q_md5_device_overlapped.txt)
        enc_stream.synchronize()
        t_enc = time.perf_counter() - t0
        tm0 = time.perf_counter()
        dev_digests = flac_amd.engine.md5_many_device(raw_all.data_ptr(), offs_b, lens_b, device=dev.index or 0, stream=enc_stream.cuda_stream)
        t_dev_md5 = time.perf_counter() - tm0
    else:
        enc_stream.synchronize()
        t_enc = time.perf_counter() - t0
    tot_h = totals.cpu().tolist()
    # the digests: the tracks' sample bytes on the host, eight tracks per pass
    t1 = time.perf_counter()
    digests = [bytes(16)] * ntracks
    if on_device:
        digests, t_prep, t_md5 = dev_digests, 0.0, t_dev_md5
    elif want_md5:
        cache = {}
        bufs = []
        for lo, hi in ranges:
            parts = []
            f = lo
            while f < hi:
                rep, b0 = divmod(f, BASE_FRAMES)
                nb = min(BASE_FRAMES - b0, hi - f)
                key = rep_params(rep)
                if key not in cache:
                    cache[key] = host_frames(base, rep * BASE_FRAMES, (rep + 1) * BASE_FRAMES)
                nsamp = min(nb * BLOCK, total_samples - f * BLOCK)
                parts.append(cache[key][b0 * BLOCK:b0 * BLOCK + nsamp])
                f += nb
            bufs.append(np.ascontiguousarray(np.concatenate(parts, axis=0)) if parts else np.zeros((0, CH), np.int16))
        t_prep = time.perf_counter() - t1
        t2 = time.perf_counter()
        if md5_threads > 1:
            # (several hashing threads, each eight chains wide: ctypes releases the GIL inside the call)
            chunks = [list(range(i, ntracks, md5_threads)) for i in range(md5_threads)]
            res = {}

            def work(idx):
                for i, d in zip(idx, md5_many([bufs[i] for i in idx])):
                    res[i] = d
            ths = [threading.Thread(target=work, args=(c,)) for c in chunks if c]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            digests = [res[i] for i in range(ntracks)]
        else:
            digests = md5_many(bufs)
        t_md5 = time.perf_counter() - t2
    else:
        t_prep = t_md5 = 0.0
    fbs = fb_all.cpu().numpy().astype(np.uint32)
    streams = []
    for t, (lo, hi) in enumerate(ranges):
        nsamp = min(hi * BLOCK, total_samples) - lo * BLOCK if hi > lo else 0
        f = fbs[lo:hi]
        header = stream_header(nsamp, int(f.min()) if f.size else 0, int(f.max()) if f.size else 0, digests[t])
        streams.append((header, out[starts[t]:starts[t] + int(tot_h[t])], f))
    return streams, {"encode_seconds": t_enc, "md5_prepare_seconds": t_prep, "md5_seconds": t_md5, "host_reads_of_totals": 1}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=0, help="encode the corpus as this many separate streams (one file per track, each with its own STREAMINFO and MD5) on one GPU")
    ap.add_argument("--md5-threads", type=int, default=0, help="--tracks: host threads hashing the tracks (0: every CPU this process may use with --input host, "
                    "one with --input device)")
    ap.add_argument("--input", choices=("host", "device"), default="host", help="--tracks: where the job finds the tracks' sample bytes -- host: in one page-locked host "
                    "buffer, as read from their files (copied to the device track by track beside the encodes, hashed by host threads straight from that buffer); "
                    "device: generated in HBM (the round-4 job: no copy, digests on the device or from a second host copy)")
    ap.add_argument("--md5", choices=("auto", "host", "device"), default="auto", help="--tracks: where the tracks' digests are computed -- device: one lane per track on the "
                    "staged sample bytes in HBM (flacgpu_md5.hip); host: AVX2, eight chains per pass, --md5-threads threads; auto: the device from 256 tracks up "
                    "(a chain is serial: 120 tracks are two wavefronts at ~30-50 MB/s per lane, 1.1-1.8 s for ten hours; 1000 tracks hash in 0.09 s -- "
                    "profiles/archive/r04_s_md5_device.txt)")
    ap.add_argument("--hours", type=float, default=10.0)
    ap.add_argument("--samples", type=int, default=0, help="corpus length in inter-channel samples (overrides --hours); a last short block is encoded as such")
    ap.add_argument("--batch-frames", type=int, default=16384)
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--gather", choices=("rccl", "host"), default="rccl")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-md5", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the process-group path with one rank too")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--check-md5", action="store_true", help="--tracks --input host: after the job (outside its clock) every track's digest is computed again with hashlib "
                    "from the input buffer and compared")
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; not under a launcher and N > 1: the job starts its N ranks itself "
                    "(flac_amd.dist.ensure_ranks); under one, N must equal its WORLD_SIZE")
    args = ap.parse_args(argv)
    from flac_amd.dist import ensure_ranks, check_world
    rank, local_rank, world = ensure_ranks(args.gpus, "flac_amd.corpus", sys.argv[1:] if argv is None else list(argv), module=True)

    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)                      # C-level banners (RCCL) must not land on the JSON line
    import torch
    import torch.distributed as dist
    import flac_amd
    from flac_amd.dist import shard_range, ordered_gather, HostShmGather

    if not torch.cuda.is_available():
        raise SystemExit("flac_amd.corpus needs a GPU: there is no CPU encode path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        check_world(args.gpus)

    total_samples = args.samples or int(round(args.hours * 3600 * RATE))
    F = (total_samples + BLOCK - 1) // BLOCK                    # 10 h: 387598 frames, the last one 2688 samples short of a block
    tail = total_samples - (F - 1) * BLOCK
    tail = 0 if tail == BLOCK else tail
    lo, hi = shard_range(F, world, rank)
    nloc = hi - lo
    base = base_clip(args.seed)
    base_dev = torch.from_numpy(base).to(dev)

    # the stream MD5: one serial chain over the whole interleaved input, on a host thread of rank 0, while the GPUs work
    md5_box = {}

    def md5_thread():
        t0 = time.perf_counter()
        h = hashlib.md5()
        cache = {}                                         # the (gain, offset) pairs repeat with period 40
        for f in range(0, F, BASE_FRAMES):
            nb = min(BASE_FRAMES, F - f)
            key = rep_params(f // BASE_FRAMES)
            if key not in cache:
                cache[key] = host_frames(base, f, f + BASE_FRAMES).tobytes()
            h.update(memoryview(cache[key])[:min(nb * BLOCK, total_samples - f * BLOCK) * CH * 2])
        md5_box["digest"] = h.digest()
        md5_box["seconds"] = time.perf_counter() - t0

    th = None
    if rank == 0 and not args.no_md5:
        th = threading.Thread(target=md5_thread)
        th.start()

    if args.tracks:
        if multi:
            raise SystemExit("--tracks runs on one GPU (tracks shard like frames do; not wired to the process group)")
        settings = flac_amd.make_settings(CH, BPS, RATE, args.level)
        ranges = track_ranges(F, args.tracks)
        eng = flac_amd.FrameEngine(settings, device=local_rank, max_batch_frames=max(hi - lo for lo, hi in ranges))
        enc_stream = torch.cuda.Stream()
        encode_tracks(eng, base, base_dev, min(F, 2 * args.tracks), min(total_samples, 2 * args.tracks * BLOCK), args.tracks, dev, enc_stream, want_md5=False)       # warm-up
        torch.cuda.synchronize()
        if args.input == "host":
            # the input, as a job that reads files would hold it when the clock starts: every track's sample bytes in page-locked host memory
            tg = time.perf_counter()
            hbuf = host_corpus(base, F, total_samples)
            t_gen = time.perf_counter() - tg
            dbufs = tracks_device_buffers(eng, F, args.tracks, dev)
            torch.cuda.synchronize()
            args.md5 = "host"
            t0 = time.perf_counter()
            streams, tm = encode_tracks_from_host(eng, hbuf, F, total_samples, args.tracks, dev, enc_stream, want_md5=not args.no_md5, md5_threads=args.md5_threads, bufs=dbufs)
            t_job = time.perf_counter() - t0
            args.md5_threads = tm["md5_threads"]
            md5_check = None
            if args.check_md5 and not args.no_md5:
                harr = hbuf.numpy()
                bad = 0
                for t, (lo, hi) in enumerate(track_ranges(F, args.tracks)):
                    want = hashlib.md5(memoryview(harr[lo * BLOCK:min(hi * BLOCK, total_samples)]).cast("B")).digest() if hi > lo else hashlib.md5(b"").digest()
                    bad += want != streams[t][0][8 + 18:8 + 34]
                md5_check = {"tracks_compared_with_hashlib": args.tracks, "digests_differing": int(bad)}
        else:
            t_gen = None
            md5_check = None
            args.md5_threads = args.md5_threads or 1
            if args.md5 == "auto":
                args.md5 = "device" if args.tracks >= 256 else "host"
            t0 = time.perf_counter()
            streams, tm = encode_tracks(eng, base, base_dev, F, total_samples, args.tracks, dev, enc_stream, want_md5=not args.no_md5, md5_threads=args.md5_threads, md5_where=args.md5)
            t_job = time.perf_counter() - t0
        import ctypes as C
        host = flac_amd.engine.load_host()
        host.flacgpu_host_check_frame_crcs.restype = C.c_int64
        host.flacgpu_host_check_frame_crcs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        bad_tracks, nbytes, t_write = 0, 0, 0.0
        for t, (header, frames, fbs) in enumerate(streams):
            data = frames.cpu().numpy()
            nbytes += data.size + len(header)
            if fbs.size and host.flacgpu_host_check_frame_crcs(data.ctypes.data, np.ascontiguousarray(fbs).ctypes.data, fbs.size, 4) != -1:
                bad_tracks += 1
            if args.out:
                tw = time.perf_counter()
                with open("%s.%04d.flac" % (args.out, t), "wb") as f:
                    f.write(header)
                    f.write(memoryview(data))
                t_write += time.perf_counter() - tw
        line = {"job": "flac -%d batch encode of a %.2f h synthetic 44.1k/16-bit stereo corpus into %d streams (one per track)" % (args.level, total_samples / RATE / 3600, args.tracks),
                "n_gpus": 1, "tracks": args.tracks, "frames": F, "samples": total_samples, "bytes": nbytes,
                "input": "page-locked host memory (generated before the clock started: %.2f s; the device buffers allocated before it, too)" % t_gen if t_gen is not None else "generated in HBM",
                "h2d_seconds": round(tm["h2d_seconds"], 4) if "h2d_seconds" in tm else None,
                "encode_seconds": round(tm["encode_seconds"], 4), "Msamples_per_s_encode": round(total_samples / tm["encode_seconds"] / 1e6, 1),
                "md5": None if args.no_md5 else "one digest per track, on the device: one lane per track over the staged sample bytes in HBM" if args.md5 == "device" else
                       "one digest per track, %s chains per register, %d host thread(s)%s" % ("16 (AVX-512) or 8 (AVX2)" if args.input == "host" else "eight", args.md5_threads,
                                                                                             ", straight from the input buffer, beside the copies and the encodes" if args.input == "host" else ""),
                "md5_seconds": round(tm["md5_seconds"], 3), "md5_prepare_seconds": round(tm["md5_prepare_seconds"], 3),
                "md5_Msamples_per_s": round(total_samples / tm["md5_seconds"] / 1e6, 1) if tm["md5_seconds"] else None,
                "job_seconds": round(t_job, 3), "tracks_with_a_bad_crc16": bad_tracks, "write_seconds": round(t_write, 3) if args.out else None,
                "md5_check": md5_check, "first_track_md5": streams[0][0][8 + 18:8 + 34].hex(), "track_md5": [h[8 + 18:8 + 34].hex() for h, _, _ in streams]}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        eng.close()
        return line

    settings = flac_amd.make_settings(CH, BPS, RATE, args.level)
    bf = min(args.batch_frames, max(nloc, 1))
    eng = flac_amd.FrameEngine(settings, device=local_rank, max_batch_frames=bf)
    cap_frame = eng.max_output_bytes(1)
    # the shard's frames land back to back; size the buffer for the measured density plus the worst case of one batch
    shard_cap = nloc * 3 * BLOCK * CH // 2 + eng.max_output_bytes(bf) if nloc else 16
    out = torch.empty(shard_cap, dtype=torch.uint8, device=dev)
    fb_all = torch.zeros(max(nloc, 1), dtype=torch.int32, device=dev)
    enc_stream = torch.cuda.Stream()
    gat = None
    if multi and args.gather == "host":
        gat = HostShmGather(F * 3 * BLOCK * CH // 2 + (1 << 20))

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: module load, RCCL channels
    if nloc:
        encode_shard(eng, base_dev, lo, min(hi, lo + min(64, nloc)), bf, dev, out, fb_all, enc_stream, F - 1, tail)
    sync()
    t0 = time.perf_counter()
    nbytes = encode_shard(eng, base_dev, lo, hi, bf, dev, out, fb_all, enc_stream, F - 1, tail) if nloc else 0
    enc_stream.synchronize()
    t_enc = time.perf_counter() - t0
    stream_t = allfb = None
    host_stream = None
    if multi and args.gather == "rccl":
        stream_t, allfb = ordered_gather(out, nbytes, fb_all[:nloc], dst=0)
    elif multi:
        off, sizes = gat.gather(out, nbytes)
        # the frame lengths are small: they still go through the collective
        _, allfb = ordered_gather(out[:0], 0, fb_all[:nloc], dst=0)
        if rank == 0:
            host_stream = gat.host[:sum(sizes)]
    else:
        stream_t, allfb = out[:nbytes], fb_all[:nloc].to(torch.int64)
    sync()
    t_job = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([t_job, t_enc], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_job, t_enc = float(t[0]), float(t[1])

    line = None
    if rank == 0:
        tw0 = time.perf_counter()
        fbs = allfb.cpu().numpy().astype(np.uint32)
        if host_stream is None:
            host_stream = stream_t.cpu()
        data = host_stream.numpy()
        t_d2h = time.perf_counter() - tw0
        assert fbs.size == F and int(fbs.sum()) == data.size
        # every frame's CRC-16 rechecked on the host (crc.c:376 over the whole frame incl. header)
        import ctypes as C
        host = flac_amd.engine.load_host()
        host.flacgpu_host_check_frame_crcs.restype = C.c_int64
        host.flacgpu_host_check_frame_crcs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        tc0 = time.perf_counter()
        bad = host.flacgpu_host_check_frame_crcs(data.ctypes.data, fbs.ctypes.data, F, min(32, os.cpu_count() or 1))
        t_crc = time.perf_counter() - tc0
        if th:
            th.join()
        digest = md5_box.get("digest", bytes(16))
        header = stream_header(total_samples, int(fbs.min()), int(fbs.max()), digest)
        t_write = None
        if args.out:
            tw = time.perf_counter()
            with open(args.out, "wb") as f:
                f.write(header)
                f.write(memoryview(data))
            t_write = time.perf_counter() - tw
        samples = total_samples
        line = {
            "job": "flac -%d batch encode of a %.2f h synthetic 44.1k/16-bit stereo corpus into one stream" % (args.level, samples / RATE / 3600),
            "n_gpus": world, "frames": F, "samples": samples, "stream_bytes": int(data.size) + len(header),
            "compressed_bytes_per_sample": round(data.size / samples, 4),
            "gather": ("ordered RCCL send/recv to rank 0" if args.gather == "rccl" else "device->shared pinned host buffer, every rank over its own PCIe link") if multi else "none (one rank, no process group)",
            "encode_gather_seconds": round(t_job, 4), "encode_seconds": round(t_enc, 4),
            "Msamples_per_s": round(samples / t_job / 1e6, 1),
            "crc16_bad_frame": int(bad), "crc16_frames_checked": F, "crc16_seconds": round(t_crc, 3),
            "md5": digest.hex(), "md5_seconds": round(md5_box.get("seconds", 0.0), 2),
            "d2h_seconds": round(t_d2h, 3), "write_seconds": None if t_write is None else round(t_write, 3),
            "min_framesize": int(fbs.min()), "max_framesize": int(fbs.max()), "out": args.out or None,
            "header_bytes": len(header),
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    elif th:
        th.join()
    eng.close()
    if gat:
        gat.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
