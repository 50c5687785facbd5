#!/usr/bin/env python3
"""bench.py -- encode throughput of the MI355X FLAC frame engine at -8, 44.1 kHz/16-bit stereo.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (flacgpu_encode_batch_device: prep -> autocorrelation -> model -> evaluation ->
pack, frames written back to back) over one batch of synthetic PCM that is already resident in HBM.  With N > 1 every
rank encodes its own contiguous frame range (weak scaling, no data-path collective) and the steps' frames go to rank 0
through the ordered RCCL gather of flac_amd/dist.py (GatherPipeline: sizes exchanged once per window of steps, the
transfers of a window overlapped with the next window's encodes; rank 0 encodes straight into the gathered stream).

Prints ONE JSON line (rank 0).  `value` = inter-channel samples encoded per second by the whole job, in M samples/s;
`roofline` prices the dominant kernel of the step (by HIP-event time) against HBM bandwidth with the ALGORITHMIC bytes
of SURVEY.md 8d (4*C bytes of PCM in + compressed bytes out per inter-channel sample); `cpu_baseline` is the
unmodified reference libFLAC (oracle/_ref, AVX2+FMA dispatch) timed on this box's host cores: one thread, one process
per core, and the library's own thread pool.  After the timed region (outside it) the frames of the last step are
checked: every CRC-16 recomputed on the host, and a sample of frames compared byte for byte with the oracle.
N = 1 also reports two side measurements with the same harness: the white-noise corpus of SURVEY.md 8d config 3
(`white_noise`) and the `flac` default preset (`level5`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RATE, BPS, CH = 44100, 16, 2
FRAMES_PER_GPU = 262144        # one GPU: 1.07 G inter-channel samples = 6.8 h of audio per step, 8.6 GB of int32 PCM resident (round 6: the
                               # kernels of a 16384-frame step are 0.2 .. 0.9 ms each and run 4 .. 8 % slower than in a long batch -- profiles/r06_w_*;
                               # 288 GB of HBM are there for this; rounds 1-5 used 16384: the line carries that figure too, `frames_16384`)
FRAMES_PER_GPU_MULTI = 65536   # N > 1 ranks: rank 0 holds two windows of every rank's frames (8 ranks x 2 x 2 steps x 0.63 GB = 20 GB)
FRAMES_SIDE = 65536            # the side measurements (other presets, white noise, 96 kHz / 24-bit)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# which kernels a phase of flacgpu_batch_phase_ms covers (the ones a workload does not launch are not in its counter pass)
KERNEL_NAMES = {"prep": "ff_kernel+prep3_kernel+prep2_kernel+prep_kernel", "autoc": "autoc3_kernel+autoc2_kernel+autoc_kernel", "model": "model_kernel",
                "eval": "evalg_kernel+evalw_kernel+eval_list_kernel+eval_kernel", "pack": "pack2_kernel+pack_plan_kernel+fo_place_kernel+pack_kernel", "scan_compact": "scan_kernel+compact_kernel"}
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md: 2.4 GHz peak engine clock.  A SIMD issues one wave64 VALU instruction per 4 cycles
VALU_ISSUE_PEAK = SIMDS * CLOCK_GHZ / 4      # G wavefront-instructions per second, the whole chip
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from the committed rocprofv3 PMC passes
UBENCH_FILE = os.path.join(ROOT, "profiles", "ubench_cycles.json")   # shader cycles per wavefront-instruction per class (scripts/ubench_cycles.hip)
# The hardware's INT32 class holds the integer ARITHMETIC (v_dot2 / v_sad / v_perm / v_mad / dpp adds: 4.1-4.4 cycles per wave64
# instruction, a quarter of the lanes per cycle; and plain adds: 2.2-2.4, half of them); shifts, logic and moves are in none of the class
# counters ("other", priced at the fast rate).  Where an ISA account of the kernel gives the split of INT32 it is used, elsewhere the
# floor is a range (all fast .. all slow).
INT32_SLOW_SHARE = {"evalg_kernel": 0.897}      # profiles/archive/r05_evalg_isa_histogram.txt: dot2 3.831 + sad 0.625 + shifted-word 0.136 of the 5.12 INT32 instructions per sample
# wavefronts per SIMD a kernel runs with (registers / LDS): which column of the microbenchmark its floor is read from
KERNEL_WAVES = {"evalg_kernel": 4, "evalw_kernel": 4, "autoc3_kernel": 2, "autoc2_kernel": 4, "prep3_kernel": 4, "pack2_kernel": 6, "ff_kernel": 4, "model_kernel": 4}


def ubench_key(w):
    """the microbenchmark's column for w wavefronts per SIMD: measured at 1, 2, 4, 5, 8 -- the largest one not above w"""
    return "w%d" % max(x for x in (1, 2, 4, 5, 8) if x <= max(1, w))


def mix_floor(name, entry):
    """What ONE kernel's instruction mix costs at best: cycles a SIMD needs per wavefront-instruction when nothing but
    the kernel's arithmetic is in the loop -- the class mix the hardware counted for it (SQ_INSTS_VALU_<class> of the committed counter
    pass) priced with the rate each class reaches alone (profiles/ubench_cycles.json: from the first start to the last end of a SIMD's
    wavefronts, at the clock measured inside the loop; HIP events agree), at eight wavefronts per SIMD and at the occupancy the kernel
    runs with.  On this chip add / shift / logic / move / fp32 issue a wave64 instruction over 2.2-2.4 cycles, everything else these
    kernels use (dot2, sad, perm, mad, add3, bfe, DPP, fp64, conversions) over 4.1-4.4; one wavefront alone issues every 4.8-5.1
    cycles, every 8.4 in a dependent chain."""
    cl = entry.get("valu_class_per_sample")
    try:
        with open(UBENCH_FILE) as fh:
            ub = json.load(fh)["classes"]
    except Exception:
        return None
    if not cl or not sum(cl.values()):
        return None
    tot = sum(cl.values())
    out = {}
    for key in ("w8", ubench_key(KERNEL_WAVES.get(name, 4))):
        c = lambda n: ub[n][key]
        fast = (c("v_add_u32") + c("v_lshrrev_b32") + c("v_xor_b32") + c("v_mov_b32")) / 4
        slow = (c("v_dot2_i32_i16") + c("v_sad_u32") + c("v_perm_b32") + c("v_mad_i32_i24")) / 4
        fixed = (cl.get("int64", 0) * c("v_mad_i64_i32") + cl.get("fma_f64", 0) * c("v_fma_f64") + cl.get("add_f64", 0) * c("v_add_f64") + cl.get("mul_f64", 0) * c("v_mul_f64")
                 + cl.get("cvt", 0) * c("v_cvt_f64_i32") + (cl.get("fma_f32", 0) + cl.get("add_f32", 0)) * c("v_fma_f32") + cl.get("other", 0) * fast)
        i32 = cl.get("int32", 0)
        lo, hi = (fixed + i32 * fast) / tot, (fixed + i32 * slow) / tot
        out[key] = lo + (hi - lo) * INT32_SLOW_SHARE[name] if name in INT32_SLOW_SHARE else (lo, hi)
    return out


def block_of(level):
    return 1152 if level < 3 else 4096


def pmc_workload(level, kind, hires):
    """the committed counter pass of a workload (profiles/pmc_traffic.json <- scripts/pmc_to_traffic.py), and whether a kernel source was
    edited since it was taken (git blob hashes recorded next to the counters)"""
    key = "hires8" if hires and level == 8 else "white8" if kind == "white" and level == 8 else "level%d" % level if kind == "music" and not hires else None
    try:
        with open(PMC_FILE) as fh:
            pmc = json.load(fh)
    except Exception:
        return None, None, None
    w = (pmc.get("workloads") or {}).get(key)
    if not w:
        return None, None, None
    import hashlib
    stale = []
    for rel, h in (pmc.get("source_hashes") or {}).items():
        try:
            data = open(os.path.join(ROOT, rel), "rb").read()
            if hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest() != h:
                stale.append(rel)
        except OSError:
            stale.append(rel)
    return w, key, stale


def synth_pcm(nframes, seed, block, kind="music"):
    """music-like 16-bit stereo (tests/signals.py: tones + coloured noise, channels correlated), or i.i.d. uniform white
    noise over the full 16-bit range"""
    import signals
    if kind == "white":
        rng = np.random.default_rng(seed)
        return rng.integers(-32768, 32768, size=(nframes * block, CH), dtype=np.int32)
    base_frames = min(nframes, 512)
    base = signals.music(base_frames * block, CH, BPS, seed=seed, rate=RATE)
    reps = (nframes + base_frames - 1) // base_frames
    if reps > 1:
        # repeat the clip with a per-repeat gain/offset so frames are not byte-identical
        parts = []
        for r in range(reps):
            g = 1.0 - 0.07 * (r % 8)
            parts.append(np.clip(np.rint(base * g) + (r % 5) - 2, -(1 << (BPS - 1)), (1 << (BPS - 1)) - 1).astype(np.int32))
        base = np.concatenate(parts, axis=0)
    return np.ascontiguousarray(base[: nframes * block])


def rank_pcm(rank, nframes, block, kind, hires):
    """the synthetic PCM rank `rank` encodes in every step (seed 1234 + rank): what measure() puts into HBM, and what rank 0
    regenerates after the timed region to check the frames it gathered from that rank"""
    if hires:
        import signals
        base_frames = min(nframes, 16384)             # (the first 16384 frames are rounds 1-5's signal; more: the clip again with gains and offsets)
        base = signals.music(base_frames * block, CH, BPS, seed=1234 + rank, rate=RATE)
        reps = (nframes + base_frames - 1) // base_frames
        if reps > 1:
            lim = 1 << (BPS - 1)
            base = np.concatenate([base if r == 0 else np.clip(np.rint(base * (1.0 - 0.07 * (r % 8))) + (r % 5) - 2, -lim, lim - 1).astype(np.int32) for r in range(reps)], axis=0)
        return np.ascontiguousarray(base[: nframes * block])
    return synth_pcm(nframes, 1234 + rank, block, kind)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the unmodified reference on this box's host cores (test infrastructure: oracle/_ref)
# ------------------------------------------------------------------------------------------------------------------
def usable_cpus():
    """CPUs this process may actually use: the visible count, cut down by the scheduler affinity and by a cgroup CPU quota
    (the GPU boxes show 256 hardware threads to a container that is allowed 16 CPUs' worth of time)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, (os.cpu_count() or 1), quota


def cpu_baseline(level, search=None):
    from oracle import pyoracle as po
    cores, visible, quota = usable_cpus()
    rate_bin = os.path.join(ROOT, "oracle", "_ref", "ref_rate")
    if not (po.have_ref() and os.path.exists(rate_bin)) or (search and any(search.values())):
        # no reference build on this box (or a search the rate program has no switch for): time the library in-process on a short clip
        pcm = synth_pcm(512 if not (search and any(search.values())) else 64, 99, block_of(level))

        def run():
            if po.have_ref():
                return po.ref_encode(pcm, BPS, RATE, level, want_bytes=False, **(search or {}))["seconds"]
            t0 = time.perf_counter()
            po.oracle_encode(pcm, BPS, RATE, level, **(search or {}))
            return time.perf_counter() - t0
        best = min(run() for _ in range(3))
        return {"value": round(pcm.shape[0] / best / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "reference" if po.have_ref() else "port",
                "sample": "%d inter-channel samples of the bench signal, flac -%d, best of 3, in-memory" % (pcm.shape[0], level), "host_cpus": visible, "usable_cpus": cores}
    # >= 10 minutes of the bench signal as a raw 16-bit file on tmpfs (input from tmpfs, output discarded)
    clip = synth_pcm(6460, 4321, 4096).astype(np.int16)                                     # 10.0 min
    nsamp = clip.shape[0]
    raw = "/dev/shm/flacgpu_bench_%d.raw" % os.getpid()
    clip.tofile(raw)
    try:
        def one(threads, reps):
            o = subprocess.check_output([rate_bin, raw, str(level), str(threads), str(reps)]).decode().split()
            return float(o[2]), float(o[3])                       # total seconds, best seconds
        _, best1 = one(1, 3)
        single = nsamp / best1 / 1e6
        # (3) of SURVEY 8d: one independent single-thread process per host CPU, three encodes each, all started together
        reps = 3
        t0 = time.perf_counter()
        procs = [subprocess.Popen([rate_bin, raw, str(level), "1", str(reps)], stdout=subprocess.PIPE) for _ in range(cores)]
        outs = [p.communicate()[0].decode().split() for p in procs]
        wall = time.perf_counter() - t0
        allcores = cores * reps * nsamp / wall / 1e6
        inside = cores * reps * nsamp / max(float(o[2]) for o in outs) / 1e6       # without process start-up / file read
        # (2): the library's own frame-parallel thread pool (flac -j N) at several thread counts, best of 3 each
        pool = {}
        for thr in sorted(set(t for t in (4, 8, 16, 32, 64, cores, 2 * cores) if 2 <= t <= max(2 * cores, 8) and t <= 128)):
            try:
                pool[thr] = round(nsamp / one(thr, 3)[1] / 1e6, 2)
            except Exception as e:                                     # a reference build without threads
                pool[thr] = str(e)
        nums = {k: v for k, v in pool.items() if isinstance(v, float)}
        best_thr = max(nums, key=nums.get) if nums else None
    finally:
        os.unlink(raw)
    return {
        "value": round(single, 3), "unit": "Msamples/s", "cores": 1, "kind": "reference",
        "sample": "%d inter-channel samples (%.1f min) of the bench signal from a raw file on tmpfs, flac -%d with MD5, output discarded, best of 3" % (nsamp, nsamp / RATE / 60, level),
        "all_cores": {"value": round(inside, 2), "value_incl_process_start": round(allcores, 2), "cores": cores,
                      "how": "%d independent single-thread processes (one per CPU this container may use: %d visible, cgroup quota %s) x %d encodes of the clip each, started together; "
                             "value = all samples / the slowest process's encode time (the better figure for the CPU)" % (cores, visible, quota, reps)},
        "library_thread_pool": {"by_threads": pool, "best_threads": best_thr, "value": nums.get(best_thr), "how": "one stream, FLAC__stream_encoder_set_num_threads, best of 3"},
        "host_cpus": visible, "usable_cpus": cores, "cgroup_cpu_quota": quota,
    }


# ------------------------------------------------------------------------------------------------------------------
# after the timed region: the frames of the last step, checked
# ------------------------------------------------------------------------------------------------------------------
def libflac_api_figures(pcm_h, nframes, block, level, d_out, total_bytes):
    """FLAC__stream_encoder_process_interleaved through libFLACgpu.so from a C client, file-less write callback (src/libFLAC/
    stream_encoder.c:2513, 2570, 3448, 3666-3686): rates, and two checks of what came out -- the frames of the whole stream equal the
    device-path step's bytes (which the line's `verified` holds to the oracle), and the first 512 frames' stream equals, byte for
    byte and metadata included, what the reference library writes for the same calls."""
    import json as _json
    import subprocess as _sp
    import tempfile as _tf
    tool = os.path.join(ROOT, "flac_amd", "lib", "api_bench")
    ref_tool = os.path.join(ROOT, "oracle", "_ref", "api_bench_ref")
    if not os.path.exists(tool):
        return {"error": "flac_amd/lib/api_bench not built"}
    n = nframes * block
    d = _tf.mkdtemp(prefix="flacgpu_api_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {}
    try:
        src = os.path.join(d, "pcm.i32")
        np.ascontiguousarray(pcm_h[:n], dtype=np.int32).tofile(src)

        def run(binary, samples, md5, streams, reps, dump=None):
            cmd = [binary, src, str(samples), "0", str(level), str(md5), str(streams), str(reps)] + ([dump] if dump else [])
            r = _sp.run(cmd, capture_output=True, text=True, timeout=600)
            if r.returncode != 0 or not r.stdout.strip():
                raise RuntimeError("%s: rc %d %s" % (os.path.basename(binary), r.returncode, r.stderr[-300:]))
            return _json.loads(r.stdout.strip().splitlines()[-1])
        dump = os.path.join(d, "gpu.flac")
        a = run(tool, n, 0, 1, 3, dump)
        stream = np.fromfile(dump, dtype=np.uint8)
        head = len(stream) - total_bytes
        frames_equal = bool(head > 0 and np.array_equal(stream[head:], d_out[:total_bytes].cpu().numpy()))
        b = run(tool, n, 1, 1, 2)
        k = max(2, min(16, usable_cpus()[0]))
        c = run(tool, n, 1, k, 2)
        out = {"one_stream_md5_off_Msamples_per_s": a["Msamples_per_s"], "one_stream_md5_on_Msamples_per_s": b["Msamples_per_s"],
               "streams_at_once": k, "streams_at_once_md5_on_Msamples_per_s": c["Msamples_per_s"], "samples_per_stream": n,
               "seconds": {"one_md5_off": a["seconds_best"], "one_md5_off_first_call_incl_runtime_start": a["seconds_first"], "one_md5_on": b["seconds_best"], "streams_md5_on": c["seconds_best"]},
               "frames_equal_the_device_step": frames_equal, "streams_identical": bool(c["streams_identical"]),
               "what": "libFLACgpu.so from one C thread per stream: FLAC__stream_encoder_init_stream (write/seek/tell callbacks to memory), process_interleaved in 1 Mi-sample calls from host memory, finish; best of the repetitions (the first includes the HIP runtime's start)"}
        prefix = min(512 * block, n)
        if os.path.exists(ref_tool):
            g, r = os.path.join(d, "g.flac"), os.path.join(d, "r.flac")
            run(tool, prefix, 1, 1, 1, g)
            rr = run(ref_tool, prefix, 1, 1, 1, r)
            out["prefix_file_equals_reference"] = bool(open(g, "rb").read() == open(r, "rb").read())
            out["reference_same_call_Msamples_per_s"] = rr["Msamples_per_s"]
        out["verified_ok"] = bool(frames_equal and out.get("prefix_file_equals_reference", False) and a["ok"] and b["ok"] and c["ok"] and c["streams_identical"])
    finally:
        import shutil as _sh
        _sh.rmtree(d, ignore_errors=True)
    return out


def verify_step(pcm_h, out_bytes, fb, first_frame, level, block, nsample=96, search=None):
    """every CRC-16 recomputed on the host; `nsample` frames (the ends of the batch, the XCD remap boundaries, random ones)
    re-encoded by the oracle with their frame number and compared byte for byte"""
    import ctypes as C
    import flac_amd
    from oracle import pyoracle as po
    host = flac_amd.engine.load_host()
    host.flacgpu_host_check_frame_crcs.restype = C.c_int64
    host.flacgpu_host_check_frame_crcs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    nframes = fb.size
    fb32 = np.ascontiguousarray(fb.astype(np.uint32))
    if int(fb32.astype(np.int64).sum()) != out_bytes.size:
        return {"crc16_frames_checked": 0, "crc16_first_bad_frame": 0, "frames_compared_with_oracle": 0, "frames_differing": [], "ok": False,
                "error": "frame lengths add up to %d bytes, the segment has %d" % (int(fb32.astype(np.int64).sum()), out_bytes.size)}
    out_bytes = np.ascontiguousarray(out_bytes)
    bad = int(host.flacgpu_host_check_frame_crcs(out_bytes.ctypes.data, fb32.ctypes.data, nframes, min(32, os.cpu_count() or 1)))
    offs = np.concatenate([[0], np.cumsum(fb32.astype(np.int64))])
    rng = np.random.default_rng(5)
    picks = {0, 1, nframes - 1, nframes - 2, nframes // 2}
    per_xcd = nframes // 8
    for x in range(1, 8):
        picks.update((x * per_xcd - 1, x * per_xcd))
    picks.update(int(v) for v in rng.integers(0, nframes, nsample))
    picks = sorted(p for p in picks if 0 <= p < nframes)
    mism = []
    for f in picks:
        want = po.oracle_encode(pcm_h[f * block:(f + 1) * block], BPS, RATE, level, first_frame=first_frame + f, **(search or {}))["data"]
        got = out_bytes[offs[f]:offs[f + 1]].tobytes()
        if got != want:
            mism.append(f)
    return {"crc16_frames_checked": nframes, "crc16_first_bad_frame": bad, "frames_compared_with_oracle": len(picks), "frames_differing": mism,
            "ok": bad == -1 and not mism}


def verify_ranks(stream, sizes, fbs, nframes, level, block, kind="music", hires=False, search=None, nsample=16):
    """The multi-rank line's check (the reference drains its frames in order, stream_encoder.c:3530-3574: so must the gather):
    `stream` holds the frames of every rank of ONE step in rank order, sizes[r] bytes from rank r, fbs[r] its frame lengths.
    Rank r's segment must be the frames r*nframes .. (r+1)*nframes - 1 of the job: every CRC-16 is recomputed, and frames of
    EVERY rank's segment (its ends, the XCD boundaries, `nsample` random ones) are encoded again by the oracle from that rank's
    PCM (rank_pcm) with their job-wide frame numbers and compared byte for byte."""
    per_rank, off = [], 0
    for r, sz in enumerate(sizes):
        seg = stream[off:off + sz]
        v = verify_step(rank_pcm(r, nframes, block, kind, hires), seg, np.asarray(fbs[r]), r * nframes, level, block, nsample=nsample, search=search)
        v["rank"] = r
        per_rank.append(v)
        off += sz
    bad = [v["rank"] for v in per_rank if not v["ok"]]
    return {"ranks_checked": len(per_rank), "ok": not bad and off == len(stream), "ranks_failing": bad,
            "crc16_frames_checked": sum(v["crc16_frames_checked"] for v in per_rank),
            "frames_compared_with_oracle": sum(v["frames_compared_with_oracle"] for v in per_rank),
            "frames_differing": [[v["rank"], f] for v in per_rank for f in v["frames_differing"]],
            "crc16_first_bad_frame": next(([v["rank"], v["crc16_first_bad_frame"]] for v in per_rank if v["crc16_first_bad_frame"] != -1), -1),
            "errors": [[v["rank"], v["error"]] for v in per_rank if "error" in v]}


def clock_probe(dev_index, run, busy_seconds=0.12):
    """The engine clock the chip holds while `run(n)` keeps it busy (the VALU yardstick assumed the 2.4 GHz peak): a
    one-wavefront kernel on a side stream times sleeps of 130 048 shader cycles against the constant 100 MHz counter
    (flacgpu_debug_clock_probe), ~16 k samples per second, beside steps of the workload; the same once more on the idle chip.
    Outside every timed region."""
    import ctypes as C
    import torch
    import flac_amd
    lib = flac_amd.engine.load_engine()
    lib.flacgpu_debug_clock_probe.restype = C.c_int
    lib.flacgpu_debug_clock_probe.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    side = torch.cuda.Stream()

    def one(busy):
        n = 1500                                    # ~90-100 ms of probing
        buf = torch.zeros(n * 4, dtype=torch.int64, device="cuda:%d" % dev_index)
        torch.cuda.synchronize()
        if lib.flacgpu_debug_clock_probe(dev_index, side.cuda_stream, n, buf.data_ptr()) != 0:
            return None
        if busy:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < busy_seconds:
                run(4)
        torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(n, 4).astype(np.float64)
        a = a[a[:, 1] > 0]
        if not len(a):
            return None
        mhz = a[:, 3] / a[:, 1] * 100.0             # nominal cycles of the sleep / its length in 100 MHz ticks
        memtime_mhz = a[:, 2] / a[:, 1] * 100.0
        # (the first and last samples may lie outside the busy stretch: the middle 80 %)
        lo, hi = len(mhz) // 10, len(mhz) - len(mhz) // 10
        m = mhz[lo:hi] if busy else mhz
        return {"mhz_mean": round(float(m.mean()), 1), "mhz_min": round(float(m.min()), 1), "mhz_p10": round(float(np.percentile(m, 10)), 1), "mhz_max": round(float(m.max()), 1),
                "samples": int(len(m)), "s_memtime_mhz_mean": round(float(memtime_mhz[lo:hi].mean()), 1)}
    busy, idle = one(True), one(False)
    if busy is None:
        return None
    busy["idle_mhz_mean"] = idle["mhz_mean"] if idle else None
    busy["how"] = "16 x s_sleep 127 = 130048 shader cycles timed against s_memrealtime (100 MHz) by one wavefront on a side stream, beside steps of this workload"
    return busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node.  Not under a launcher and N > 1: bench.py starts its N ranks itself "
                    "(flac_amd.dist.ensure_ranks -> torch.distributed.run); under one, N must equal its WORLD_SIZE (else exit code 3)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default: %d on one GPU, %d per rank on several)" % (FRAMES_PER_GPU, FRAMES_PER_GPU_MULTI))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the white-noise and -5 side measurements")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the last step's frames (outside the timed region)")
    ap.add_argument("--timing-every", type=int, default=1, help="per-kernel HIP-event timing on every n-th step of the timed region (1: every step)")
    ap.add_argument("--no-api", action="store_true", help="skip the libflac_api side figures (the libFLAC encoder API driven by a C client, outside the timed region)")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode_only side figure (the step's output decoded again as a bare stream, outside the timed region)")
    ap.add_argument("--hires", action="store_true", help="96 kHz / 24-bit stereo (BASELINE.json config 4): a side measurement")
    ap.add_argument("--white", action="store_true", help="white-noise corpus as the main measurement (side measurement)")
    ap.add_argument("--exhaustive", action="store_true", help="flac -8e: not the headline workload, a side measurement")
    ap.add_argument("--prec-search", action="store_true", help="flac -8p")
    ap.add_argument("--level", type=int, default=8, help="compression preset -0..-8 (the metric is quoted at -8: other levels are side measurements; "
                    "-0..-2 use the preset's 1152-sample blocks)")
    ap.add_argument("--window", type=int, default=4, help="multi-rank: steps per gather window")
    ap.add_argument("--gather", choices=("rccl", "hostshm", "none"), default="rccl", help="multi-rank: how a step's frames reach their destination -- rccl: point-to-point over xGMI "
                    "into rank 0's HBM (the north star's gather); hostshm: every rank copies over its own PCIe link into one shared pinned host buffer; "
                    "none: they stay where they were encoded (encode-only scaling: what the funnel costs is this line against the rccl one)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank pipeline (process group, windowed ordered gather) even with one rank")
    ap.add_argument("--no-clock", action="store_true", help="skip the engine-clock probe (outside the timed region)")
    ap.add_argument("--no-side-gathers", action="store_true", help="multi-rank rccl line: skip the encode-only (--gather none) and hostshm figures measured beside it in the same run")
    ap.add_argument("--side-timeout", type=int, default=420, help="multi-rank rccl line: seconds the side figures may take before rank 0 prints the main line without them and exits 1")
    args = ap.parse_args()
    # one call, N workers: `python bench.py --gpus 8` starts its eight ranks itself (and never prints n_gpus: 8 from fewer)
    from flac_amd.dist import ensure_ranks, check_world
    launched_by = "a launcher (WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else "bench.py itself"
    if "WORLD_SIZE" not in os.environ and ((args.gpus or 1) > 1 or (args.gpus and os.environ.get("FLAC_AMD_FORCE_LAUNCH") == "1")):
        os.environ["FLACGPU_BENCH_SELF_LAUNCHED"] = "1"
    elif os.environ.get("FLACGPU_BENCH_SELF_LAUNCHED") == "1":
        launched_by = "bench.py --gpus %d (flac_amd.dist.ensure_ranks -> torch.distributed.run)" % (args.gpus or 1)
    rank, local_rank, world = ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    global RATE, BPS
    if args.hires:
        RATE, BPS = 96000, 24
    LEVEL = args.level
    BLOCK = block_of(LEVEL)

    # stdout must carry exactly one JSON line: libraries underneath (RCCL prints a version banner from C) get stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import flac_amd
    from flac_amd.dist import GatherPipeline, HostShmPipeline

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d wants device %d, %d visible\n" % (rank, local_rank, torch.cuda.device_count()))
        raise SystemExit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        check_world(args.gpus)

    nframes_main = args.frames if args.frames else (FRAMES_PER_GPU_MULTI if (world > 1 or args.force_dist) else FRAMES_PER_GPU)
    search = dict(exhaustive=int(args.exhaustive), prec_search=int(args.prec_search))

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(level, kind, steps, warmup, use_dist, gather=None, window=None, frames=None):
        """K timed steps of one configuration; returns the numbers and what the verification needs"""
        nframes = frames or nframes_main
        gather = gather or args.gather
        window = window or args.window
        block = block_of(level)
        settings = flac_amd.make_settings(CH, BPS, RATE, level, **search)
        eng = flac_amd.FrameEngine(settings, device=local_rank, max_batch_frames=nframes)
        # phase timing (HIP events on the engine's stream around its kernels) on every --timing-every-th step: instrumentation that
        # costs the stream ~4.6 us per record; the per-kernel durations of the line are the averages over the steps that carry it
        eng.set_phase_timing(args.timing_every)
        pcm_h = rank_pcm(rank, nframes, block, kind, args.hires)
        d_pcm = torch.from_numpy(pcm_h).to(dev)
        cap = eng.max_output_bytes(nframes)
        first_frame = rank * nframes
        if not use_dist or gather == "none":
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_fb = torch.empty(nframes, dtype=torch.int32, device=dev)
            d_total = torch.zeros(1, dtype=torch.int64, device=dev)
            enc_stream = torch.cuda.Stream()
            gp = None

            def run(nsteps):
                for _ in range(nsteps):
                    eng.encode_device(d_pcm.data_ptr(), nframes, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(),
                                      first_frame_number=first_frame, stream=enc_stream.cuda_stream)
        else:
            if gather == "hostshm":
                # room per rank and step in the shared host buffer: what one step of this signal really takes, plus a quarter
                # (the worst case, every frame VERBATIM, would pin 2 windows x world x 288 MB per step)
                d_probe = torch.empty(cap, dtype=torch.uint8, device=dev)
                d_pfb = torch.empty(nframes, dtype=torch.int32, device=dev)
                d_ptot = torch.zeros(1, dtype=torch.int64, device=dev)
                eng.encode_device(d_pcm.data_ptr(), nframes, d_probe.data_ptr(), cap, d_pfb.data_ptr(), d_ptot.data_ptr(), first_frame_number=first_frame, stream=0)
                torch.cuda.synchronize()
                hcap = (int(d_ptot.item()) * 5 // 4 + 4095) & ~4095
                del d_probe, d_pfb, d_ptot
                gp = HostShmPipeline(cap, nframes, dev, window=window, host_cap_bytes=hcap)
            else:
                gp = GatherPipeline(cap, nframes, dev, window=window)
            state = {"k": 0}

            def run(nsteps):
                k0 = state["k"]
                for k in range(k0, k0 + nsteps):
                    gp.wait_slot_free(k)
                    out, fbt, tot = gp.slot(k)
                    eng.encode_device(d_pcm.data_ptr(), nframes, out.data_ptr(), cap, fbt.data_ptr(), tot.data_ptr(),
                                      first_frame_number=first_frame, stream=gp.enc_stream.cuda_stream)
                    gp.step_done(k)
                gp.flush()
                # the next call starts on a window boundary
                state["k"] = ((k0 + nsteps + gp.K - 1) // gp.K) * gp.K

        if warmup:
            run(warmup)
        sync()
        nwin_warm = len(gp.win_log) if gp is not None else 0
        t0 = time.perf_counter()
        run(steps)
        sync()
        elapsed = time.perf_counter() - t0
        # per-kernel durations of the timed steps: HIP events the engine recorded on its stream around every launch (it keeps
        # the sets of its last 64 batches, so nothing had to sync inside the timed region)
        phase_ms = [ph for ph in (eng.last_phase_ms(back) for back in range(min(steps, 64))) if ph is not None]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        res = {"elapsed": elapsed, "steps": steps, "block": block, "level": level, "kind": kind, "nframes": nframes}
        if rank == 0 and gp is None and not use_dist and not args.no_clock:
            try:
                res["clock"] = clock_probe(local_rank, run)
            except Exception as e:                     # (a development aid must not take the line down; before the verify side measurement,
                                                       #  behind which the pack kernel also writes the verify pass's hints)
                res["clock"] = {"error": str(e)}
        if rank == 0 and gp is None and not use_dist and not args.no_verify:
            # side measurement, outside the timed region: the whole batch decoded again on the device and compared with its input
            # (flacgpu_verify_batch_device: what set_verify(true) costs per batch)
            d_res = torch.zeros(32, dtype=torch.uint8, device=dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            with torch.cuda.stream(enc_stream):
                eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nframes, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=first_frame, stream=enc_stream.cuda_stream)
                ev0.record(enc_stream)
                for _ in range(reps):
                    eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nframes, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=first_frame, stream=enc_stream.cuda_stream)
                ev1.record(enc_stream)
            enc_stream.synchronize()
            seq_ms = ev0.elapsed_time(ev1) / reps          # no hints yet (the verify buffers did not exist when the batch was packed): a lane per frame
            v_seq = flac_amd.VerifyResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
            # what set_verify(true) runs: the batch is packed again, now leaving its run starts behind for the verify pass
            with torch.cuda.stream(enc_stream):
                run(1)
                ev0.record(enc_stream)
                for _ in range(reps):
                    eng.verify_device(d_out.data_ptr(), d_fb.data_ptr(), nframes, d_pcm.data_ptr(), d_res.data_ptr(), first_frame_number=first_frame, stream=enc_stream.cuda_stream)
                ev1.record(enc_stream)
            enc_stream.synchronize()
            vms = ev0.elapsed_time(ev1) / reps
            v = flac_amd.VerifyResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
            res["device_verify"] = {"status": int(v.status) | int(v_seq.status), "frames_decoded_and_compared": nframes, "ms_per_batch": round(vms, 4),
                                    "frames_verified_a_thread_per_run": eng.verify_hinted_frames(),
                                    "ms_per_batch_lane_per_frame": round(seq_ms, 4),
                                    "decode_Msamples_per_s": round(nframes * block / vms / 1e3, 1),
                                    "encode_plus_verify_Msamples_per_s": round(nframes * block / (vms + elapsed / steps * 1e3) / 1e3, 1)}
        if rank == 0 and gp is None and not use_dist and not args.no_verify and not args.no_decode:
            # side measurement, outside the timed region: the step's output taken as a stream nobody vouches for --
            # no frame lengths, no input to compare with -- and decoded on the device: sync-code scan, a lane per frame, CRC-16,
            # interleaved PCM in HBM (flacgpu_decode_stream_device); then compared with the step's input
            try:
                from flac_amd.stream_decoder import StreamDecoder, StreamInfo
                sdec = StreamDecoder(local_rank)
                tb = int(d_total.item())
                info = StreamInfo(1, block, block, RATE, CH, BPS)
                d_dec = torch.empty(nframes * block * CH, dtype=torch.int32, device=dev)
                rc, r0, ev = sdec.decode_device(d_out.data_ptr(), tb, 0, info, d_dec.data_ptr(), d_dec.numel())      # (buffers grow on this one)
                torch.cuda.synchronize()
                walls, devms = [], []
                for _ in range(3):
                    t1 = time.perf_counter()
                    rc, r0, ev = sdec.decode_device(d_out.data_ptr(), tb, 0, info, d_dec.data_ptr(), d_dec.numel())
                    torch.cuda.synchronize()
                    walls.append((time.perf_counter() - t1) * 1e3)
                    devms.append({"scan": r0.ms_scan, "decode": r0.ms_decode, "place": r0.ms_place, "total": r0.ms_total})
                best = min(range(3), key=lambda k: walls[k])
                same = bool(torch.equal(d_dec.view(-1), d_pcm.view(-1)))
                res["decode_only"] = {"rc": int(rc), "samples": int(r0.samples), "frames": int(r0.frames), "errors": int(r0.nevents), "sync_codes": int(r0.candidates),
                                      "redecoded_frames": int(r0.redecoded_frames), "pcm_equals_input": same,
                                      "ms_wall": round(walls[best], 3), "ms_device": {k: round(v, 3) for k, v in devms[best].items()},
                                      "Msamples_per_s": round(nframes * block / walls[best] / 1e3, 1), "stream_bytes": tb,
                                      "what": "flacgpu_decode_stream_device on the step's %d bytes as a bare stream: frames found by sync code, nothing known but STREAMINFO's fields; wall time of the call (two host reads of the candidate table inside)" % tb}
                sdec.close()
                del d_dec
            except Exception as e:
                res["decode_only"] = {"error": repr(e)}
        if rank == 0 and gp is None and not use_dist and not args.no_verify and not args.no_api and level == LEVEL and kind == main_kind and not args.hires:
            # side measurement, outside the timed region: the boundary north_star names -- FLAC__stream_encoder_* --
            # driven by a C client (flac_amd/lib/api_bench <- flac_amd/csrc/tools/api_bench.c) on this step's PCM from host memory:
            # one stream with MD5 off and on (the API's default), and K streams at once with MD5 on
            try:
                api_frames = min(nframes, 16384)                   # (67 M samples per stream, as in round 5: sixteen streams of the whole step would be 137 GB of host buffers)
                res["libflac_api"] = libflac_api_figures(pcm_h, api_frames, block, level, d_out, int(d_fb[:api_frames].to(torch.int64).sum().item()))
            except Exception as e:
                res["libflac_api"] = {"error": repr(e)}
        verified = None
        if use_dist and not args.no_verify:
            # the multi-rank check, outside the timed region: EVERY rank's frames of the last step
            if gp is None:
                # --gather none: nothing was moved; every rank checks its own frames where they lie and rank 0 collects the verdicts
                tb = int(d_total.item())
                mine = verify_step(pcm_h, d_out[:tb].cpu().numpy(), d_fb.cpu().numpy(), first_frame, level, block, nsample=16, search=search)
                flags = torch.tensor([1 if mine["ok"] else 0, mine["crc16_frames_checked"], mine["frames_compared_with_oracle"]], dtype=torch.int64, device=dev)
                table = torch.empty(world * 3, dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(table, flags)
                table = table.cpu().view(world, 3)
                verified = {"ranks_checked": world, "ok": bool(table[:, 0].min().item() == 1), "ranks_failing": [r for r in range(world) if table[r, 0].item() != 1],
                            "crc16_frames_checked": int(table[:, 1].sum()), "frames_compared_with_oracle": int(table[:, 2].sum()),
                            "how": "no gather: every rank checked its own frames (CRC-16 of all, oracle on a sample, frame numbers rank * frames + f)"}
            else:
                k_last = state["k"] - 1 if (steps % gp.K == 0) else (state["k"] - gp.K + steps % gp.K - 1)
                if gather == "hostshm":
                    # (the shared host buffer holds every rank's bytes, but a rank knows only its own frame lengths: they are collected here)
                    _, _, myfb = gp.gathered(k_last)
                    allfb = torch.empty(world * nframes, dtype=torch.int32, device=dev)
                    dist.all_gather_into_tensor(allfb, myfb.contiguous().to(torch.int32))
                if rank == 0:
                    stream, sizes, fbs = gp.gathered(k_last)
                    if gather == "hostshm":
                        fbs = allfb.view(world, nframes)
                    verified = verify_ranks(stream[:sum(sizes)].cpu().numpy(), sizes, fbs.cpu().numpy(), nframes, level, block, kind, args.hires, search)
        if rank == 0:
            if gp is None:
                total_bytes = int(d_total.item())
                out_h = d_out[:total_bytes].cpu().numpy()
                fb_h = d_fb.cpu().numpy()
            else:
                k_last = state["k"] - 1 if (steps % gp.K == 0) else (state["k"] - gp.K + steps % gp.K - 1)
                stream, sizes, fbs = gp.gathered(k_last)
                total_bytes = sizes[0]
                out_h = stream[:total_bytes].cpu().numpy()            # rank 0's own frames: the head of the gathered stream
                fb_h = (fbs[0] if fbs.dim() == 2 else fbs).cpu().numpy()
                res["gathered_bytes_last_step"] = int(sum(sizes))
                res["host_syncs"] = gp.host_syncs
                # the windows of the timed steps, as rank 0 saw them: bytes it received (rccl) or copied out (hostshm) and the time
                # their transfers took on its communication stream -- next to ms_per_step this says whether the gather or the
                # encode set the pace
                res["gather_windows"] = gp.window_stats()[nwin_warm:]
                res["gather_backend"] = gp.backend
            kms = {k: float(np.mean([ph[k] for ph in phase_ms])) for k in phase_ms[0]} if phase_ms else {k: 0.0 for k in ("prep", "autoc", "model", "eval", "pack", "scan_compact")}
            samples_per_step = nframes * block
            out_bps = total_bytes / samples_per_step
            alg_bytes = samples_per_step * (4 * CH + out_bps)         # PCM read once + frames written once
            dom = max(kms, key=kms.get)
            achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9 if kms[dom] else 0.0          # (--timing-every 0: no kernel times)
            traffic = valu_busy = traffic_step = None
            valu = {}
            pw, pkey, pstale = pmc_workload(level, kind, args.hires)
            if pw and pw.get("blocksize") == block and not any(search.values()):
                # counter-derived figures: the committed pass of THIS workload (bytes and instructions per frame scale with the batch)
                K = pw["kernels"]
                for name in KERNEL_NAMES[dom].split("+"):
                    if name in K:
                        traffic = (traffic or 0) + int(K[name]["hbm_bytes_per_frame"] * nframes)
                        valu_busy = max(valu_busy or 0.0, K[name].get("valu_busy_frac", 0.0))
                step_kernels = set(n for names in KERNEL_NAMES.values() for n in names.split("+"))
                traffic_step = int(sum(v["hbm_bytes_per_frame"] for k, v in K.items() if k in step_kernels) * nframes)
                for ph, names in KERNEL_NAMES.items():
                    ips = sum(K[nm].get("valu_wave_insts_per_sample", 0.0) for nm in names.split("+") if nm in K)
                    if ips and kms.get(ph):
                        ach = ips * samples_per_step / (kms[ph] * 1e-3) / 1e9
                        valu[ph] = {"kernel": "+".join(nm for nm in names.split("+") if nm in K), "wave_insts_per_sample": round(ips, 4), "ms": round(kms[ph], 4),
                                    "achieved_Ginst_per_s": round(ach, 1), "frac_of_issue_peak": round(ach / VALU_ISSUE_PEAK, 4)}
            res.update(value=world * samples_per_step * steps / elapsed / 1e6, ms_per_step=elapsed / steps * 1e3, kernel_ms=kms, out_bps=out_bps,
                       roofline={"bound": "hbm", "kernel": ("+".join(nm for nm in KERNEL_NAMES[dom].split("+") if pw and nm in pw["kernels"]) or KERNEL_NAMES[dom]),
                                 "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "algorithmic_bytes_per_launch": int(alg_bytes),
                                 "whole_step_frac": round(alg_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, 6),
                                 # every kernel of the step, not only the dominant one: HBM-side bytes of the committed counter pass and
                                 # how many times the algorithmic bytes that is (the wasted re-reads between the kernels)
                                 "traffic_whole_step": traffic_step,
                                 "traffic_whole_step_over_algorithmic": round(traffic_step / alg_bytes, 3) if traffic_step else None,
                                 # what actually bounds this kernel: the share of its cycles in which it issues VALU work (committed PMC pass)
                                 "valu_busy_frac_of_committed_pmc_pass": valu_busy,
                                 # which counter pass, and the kernel sources edited since it was taken (none: the counters are this tree's)
                                 "pmc_workload": pkey, "stale": bool(pstale) if pkey else None, "stale_sources": pstale if pstale else None})
            if valu:
                tot_i = sum(v["wave_insts_per_sample"] for v in valu.values())
                res["roofline_valu"] = {"bound": "valu issue", "peak": VALU_ISSUE_PEAK, "unit": "G wavefront-instructions/s",
                                        "peak_is": "%d SIMDs x %.1f GHz / 4 cycles per wave64 instruction" % (SIMDS, CLOCK_GHZ),
                                        "per_kernel": valu, "wave_insts_per_sample_whole_step": round(tot_i, 3),
                                        "whole_step_frac_of_issue_peak": round(tot_i * samples_per_step / (elapsed / steps) / 1e9 / VALU_ISSUE_PEAK, 4),
                                        "source": "SQ_INSTS_VALU of the committed counter pass (profiles/pmc_traffic.json) x this run's HIP-event kernel times"}
                ck = res.get("clock") or {}
                if ck.get("mhz_mean"):
                    # the same against the clock the chip really held under this workload (clock_probe): the peak above assumes 2.4 GHz
                    peak_m = SIMDS * ck["mhz_mean"] / 1000.0 / 4
                    # ... and against what each kernel's own instruction mix costs at best (mix_floor): cycles per wavefront-instruction
                    # achieved at the measured clock, next to the mix's floor at eight wavefronts per SIMD and at the kernel's occupancy
                    mf = {}
                    for ph, v in valu.items():
                        names = [nm for nm in v["kernel"].split("+") if nm in K and K[nm].get("valu_class_per_sample")]
                        if not names:
                            continue
                        nm = max(names, key=lambda x: K[x].get("valu_wave_insts_per_sample", 0.0))
                        fl = mix_floor(nm, K[nm])
                        if not fl:
                            continue
                        ach_c = SIMDS * ck["mhz_mean"] * 1e6 * (v["ms"] * 1e-3) / (v["wave_insts_per_sample"] * samples_per_step)
                        wk = ubench_key(KERNEL_WAVES.get(nm, 4))
                        rnd = lambda x: [round(x[0], 3), round(x[1], 3)] if isinstance(x, tuple) else round(x, 3)
                        frac = lambda x: [round(x[0] / ach_c, 3), round(min(1.0, x[1] / ach_c), 3)] if isinstance(x, tuple) else round(x / ach_c, 3)
                        mf[ph] = {"kernel": nm, "cycles_per_wave_instruction": round(ach_c, 3), "waves_per_simd": KERNEL_WAVES.get(nm, 4),
                                  "floor_at_8_waves": rnd(fl["w8"]), "floor_at_this_occupancy": rnd(fl[wk]),
                                  "frac_of_floor_at_8_waves": frac(fl["w8"]), "frac_of_floor_at_this_occupancy": frac(fl[wk]),
                                  "classes_per_sample": K[nm]["valu_class_per_sample"]}
                    if mf:
                        res["roofline_valu"]["mix_floor"] = dict(mf, how="floor = sum over the hardware's VALU classes (SQ_INSTS_VALU_<class> of the committed counter pass) of share x the cycles "
                                                                 "per wavefront-instruction the class reaches alone (profiles/ubench_cycles.json: first start to last end of a SIMD's W wavefronts, clock "
                                                                 "measured inside the loop); the INT32 class holds the 2.3-cycle plain adds and the 4.2-cycle dot2 / sad / perm / mad: [all fast, all slow] "
                                                                 "unless an ISA account gives the split (evalg_kernel); shifts, logic and moves ('other') issue in 2.3 cycles, fp64 and conversions in 4.1-4.5")
                    res["roofline_valu"]["at_measured_clock"] = {
                        "clock_mhz": ck["mhz_mean"], "peak": round(peak_m, 1),
                        "whole_step_frac_of_issue_peak": round(tot_i * samples_per_step / (elapsed / steps) / 1e9 / peak_m, 4),
                        "per_kernel_frac": {k: round(v["achieved_Ginst_per_s"] / peak_m, 4) for k, v in valu.items()},
                        "cycles_per_wave_instruction_whole_step": round(SIMDS * ck["mhz_mean"] * 1e6 * (elapsed / steps) / (tot_i * samples_per_step), 3),
                        "note": "4 cycles per wave64 instruction is the yardstick of rounds 3-5: scripts/ubench_cycles.hip (round 6, cycles counted inside the kernel, HIP events "
                                "agree) measures 4.1-4.4 for dot2 / sad / perm / mad / fp64 and 2.2-2.4 for add / shift / logic / move -- a kernel's own floor is in mix_floor"}
            if verified is not None:
                res["verified"] = verified
            elif not args.no_verify:
                res["verified"] = verify_step(pcm_h, out_h, fb_h, first_frame, level, block, search=search)
        if rank == 0:
            try:
                # frames that gave up waiting in the fused output (placed from their slots by fo_place_kernel): time, never bytes
                tot_fb, max_fb = eng.fused_fallbacks()
                res["fused_output_fallback_frames"] = {"since_the_engine_was_created": tot_fb, "most_in_one_batch": max_fb}
            except Exception as e:
                res["fused_output_fallback_frames"] = {"error": str(e)}
        if gp is not None and hasattr(gp, "close"):
            gp.close()
        eng.close()
        del d_pcm
        torch.cuda.empty_cache()
        return res

    main_kind = "white" if args.white else "music"
    m = measure(LEVEL, main_kind, args.steps, args.warmup, multi)
    extras = {}
    side = {}
    def compose(extras, side):
        """the bench line from the main measurement m (rank 0)"""
        sig = "white noise" if args.white else "music-like synthetic PCM"
        line = {
            "metric": ("encode Msamples/s at -8, 44.1k/16-bit stereo; bit-exact vs libFLAC" if not args.hires else "encode Msamples/s at -8, 96k/24-bit stereo (side measurement)")
                      if LEVEL == 8 else "encode Msamples/s at -%d (side measurement; the metric is quoted at -8)" % LEVEL,
            "value": round(m["value"], 3), "unit": "Msamples/s", "n_gpus": world, "launched_by": launched_by, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": "flac -%d%s%s (%s) on %s stereo, "
                                   "%d frames x %d samples per GPU per step, %s resident in HBM" % (LEVEL, "e" if args.exhaustive else "", "p" if args.prec_search else "",
                                   "max LPC order 12, subdivide_tukey(3), mid/side, partition order <= 6" if LEVEL == 8 else "the preset's settings, stream_encoder.c:117-140",
                                   "96k/24-bit" if args.hires else "44.1k/16-bit", nframes_main, m["block"], sig),
                       "frames_per_gpu_per_step": nframes_main, "blocksize": m["block"], "channels": CH, "bits_per_sample": BPS,
                       "samples_are": "inter-channel (x2 for channel-samples)",
                       "input": "every step encodes the SAME resident batch again (%.0f MB of int32 PCM per GPU: larger than the 256 MB Infinity Cache, so it streams from "
                                "HBM every step, but it is one buffer, not a corpus)" % (nframes_main * m["block"] * CH * 4 / 1e6),
                       "parallelism": ("frame-shard x%d, no gather: every rank's frames stay in its HBM (encode-only scaling; the difference to the rccl line is the funnel)" % world
                                       if args.gather == "none" else
                                       "frame-shard x%d + ordered RCCL gather of every step's frames to rank 0 (sizes exchanged once per window of %d steps, "
                                       "transfers overlapped with the next window's encodes, rank 0 encodes in place)" % (world, args.window) if args.gather == "rccl" else
                                       "frame-shard x%d + every rank copies each step's frames over its own PCIe link into one shared pinned host buffer at the scanned offset "
                                       "(sizes exchanged once per window of %d steps)" % (world, args.window)) if multi else "one GPU, no process group",
                       "compressed_bytes_per_sample": round(m["out_bps"], 4)},
            "kernel_ms": {k: round(v, 4) for k, v in m["kernel_ms"].items()},
            "clock": m.get("clock"),
            "fused_output_fallback_frames": m.get("fused_output_fallback_frames"),
            "roofline": dict(m["roofline"], note="-8 is VALU bound (~1e3 integer+fp64 ops per sample; the dominant kernels issue VALU work ~80% of their cycles, "
                                                 "profiles/*pmc*); the HBM fraction is reported because the north star asks for it; traffic = (2*FETCH_SIZE+WRITE_SIZE) "
                                                 "of the committed PMC pass scaled to this batch; whole_step_frac prices the whole step instead of its dominant kernel"),
        }
        if "roofline_valu" in m:
            line["roofline_valu"] = m["roofline_valu"]
        if "verified" in m:
            line["verified_frames"] = m["verified"]["frames_compared_with_oracle"]
            line["verified"] = m["verified"]
        if "device_verify" in m:
            line["device_verify"] = m["device_verify"]
        if "decode_only" in m:
            line["decode_only"] = m["decode_only"]
        if "libflac_api" in m:
            line["libflac_api"] = m["libflac_api"]
        if multi:
            wins = m.get("gather_windows") or []
            wms = [w["ms"] for w in wins if w["ms"] is not None]
            line["gather"] = {"mode": args.gather, "backend": m.get("gather_backend"), "world_size_seen": dist.get_world_size(), "window": args.window,
                              "bytes_gathered_last_step": m.get("gathered_bytes_last_step"), "host_reads_of_sizes": m.get("host_syncs"),
                              "windows": wins,
                              "rank0_transfer_ms_per_step": round(sum(wms) / max(1, sum(w["steps"] for w in wins)), 4) if wms else None,
                              "rank0_transfer_GBps": round(sum(w["bytes"] for w in wins) / (sum(wms) * 1e-3) / 1e9, 2) if wms and sum(wms) > 0 else None,
                              "note": "ms = time a window's transfers occupied rank 0's communication stream (they run beside the next window's encodes); "
                                      "if rank0_transfer_ms_per_step approaches ms_per_step the gather sets the pace"}
        line.update(extras)
        line.update(side)
        return line

    def side_figures():
        """multi-rank rccl line: the encode-only and the shared-host-buffer figures of the same run (every rank calls this)"""
        side = {}
        side_steps = max(args.window, args.steps // 2)
        eo = measure(LEVEL, main_kind, side_steps, 1, True, "none")
        fault = os.environ.get("FLACGPU_BENCH_SIDE_FAULT")        # (tests/test_bench_gpu.py: what the line looks like when these figures fail)
        if fault == "raise":
            raise RuntimeError("injected fault")
        if fault == "hang":
            import threading
            threading.Event().wait()
        # the shared host buffer lives in /dev/shm and every rank page-locks all of it: windows of two steps for this side figure
        # (eight ranks x four slots x 8 x 240 MB = 7.7 GB instead of 15); rank 0 looks whether they fit and tells the others
        hs_window = min(args.window, 2)
        need = 2 * hs_window * world * ((int(m.get("gathered_bytes_last_step") or 0) // max(1, world)) * 5 // 4 + 4096) if rank == 0 else 0
        verdict = [None]
        if rank == 0:
            try:
                st = os.statvfs("/dev/shm")
                free = st.f_bavail * st.f_frsize
                verdict[0] = None if need + (256 << 20) <= free else "the windows need %d MB of /dev/shm, %d MB free" % (need >> 20, free >> 20)
            except OSError as e:
                verdict[0] = "no /dev/shm: %s" % e
        dist.broadcast_object_list(verdict, src=0)
        hs = measure(LEVEL, main_kind, side_steps, 1, True, "hostshm", window=hs_window) if verdict[0] is None else None
        if rank == 0:
            def side_line(r, how):
                wins = r.get("gather_windows") or []
                wms = [w["ms"] for w in wins if w["ms"] is not None]
                return {"value": round(r["value"], 3), "unit": "Msamples/s", "ms_per_step": round(r["ms_per_step"], 4), "steps": r["steps"], "how": how,
                        "verified": {k: r["verified"][k] for k in ("ranks_checked", "ok", "ranks_failing", "crc16_frames_checked", "frames_compared_with_oracle")} if r.get("verified") else None,
                        "rank0_transfer_ms_per_step": round(sum(wms) / max(1, sum(w["steps"] for w in wins)), 4) if wms else None}
            side["encode_only"] = side_line(eo, "--gather none in the same run: every rank's frames stay in its HBM; rccl value / this = what the funnel into rank 0 costs")
            side["hostshm"] = side_line(hs, "--gather hostshm in the same run: every rank copies its frames over its own PCIe link into one shared pinned host buffer") \
                if hs is not None else {"skipped": verdict[0]}
        return side

    emitted = [False]

    def emit(line):
        if not emitted[0]:
            emitted[0] = True
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    def bail(why):
        """rank 0, multi-rank: the side figures failed, hung or another rank died -- the main measurement is complete and checked:
        print its line (saying what happened to the side figures) and leave; the launcher takes the other ranks down"""
        emit(compose({}, {"side_figures": {"error": why}}))
        os._exit(1)

    if multi and args.gather == "rccl" and not args.no_side_gathers:
        # the same job twice more in the same run, so that a sub-linear rccl figure splits into encode scaling and funnel without
        # a second command: the frames stay where they were encoded (`encode_only`), and every rank copies them over its own PCIe
        # link into one shared pinned host buffer (`hostshm`).  Every rank's frames are checked in both.
        # (the main line must not depend on them: if rank 0 fails here, or another rank dies and the launcher signals this one, or
        #  nothing moves for --side-timeout seconds, rank 0 prints the main line with the reason and exits non-zero)
        from flac_amd.dist import Watchdog
        import contextlib
        with (Watchdog(args.side_timeout, bail, "the encode-only / hostshm figures") if rank == 0 else contextlib.nullcontext()):
            try:
                side = side_figures()
            except Exception as e:                   # (other ranks: the exception ends the process, the launcher ends the job)
                if rank == 0:
                    bail("%s: %s" % (type(e).__name__, e))
                raise
    if world == 1 and not multi and not args.no_extras and LEVEL == 8 and not args.hires and not args.white and not any(search.values()):
        # (one GPU, no collectives: a side measurement that fails says so in the line instead of taking the main measurement with it)
        try:
            # (the first tens of milliseconds behind an idle stretch run 3 % (-8) to 11 % (-5) below the steady state, profiles/r06_w_*:
            #  three warm-up steps and at least 60 ms of timed steps for every side figure)
            side_steps = max(6, args.steps)
            fside = min(FRAMES_SIDE, nframes_main)
            w = measure(8, "white", side_steps, 3, False, frames=fside)
            l5 = measure(5, "music", 2 * side_steps, 3, False, frames=fside)
            l0 = measure(0, "music", 4 * side_steps, 3, False, frames=min(4 * fside, nframes_main))       # (1152-sample blocks: four times the frames are the same samples)
            # the step of rounds 1-5 (16384 frames = 67 M samples, 2 ms at -8): the same kernels in short launches, at that round's step count
            small = measure(8, "music", 20, 2, False, frames=16384) if nframes_main > 16384 else None
            args.hires = True
            RATE, BPS = 96000, 24
            hr = measure(8, "music", side_steps, 3, False, frames=fside)
            args.hires = False
            RATE, BPS = 44100, 16
            if rank == 0:
                if small is not None:
                    extras["frames_16384"] = {"what": "the main measurement's workload in steps of 16384 frames (the step of rounds 1-5: BENCH_r01..r05 are this figure)",
                                              "value": round(small["value"], 3), "unit": "Msamples/s", "ms_per_step": round(small["ms_per_step"], 4), "steps": small["steps"],
                                              "kernel_ms": {k: round(v, 4) for k, v in small["kernel_ms"].items()}, "verified_ok": small.get("verified", {}).get("ok")}
                for key, r, what in (("white_noise", w, "flac -8 on i.i.d. uniform 16-bit stereo white noise (SURVEY.md 8d config 3 (i)): the 32-bit side-channel path, the largest frames"),
                                     ("level5", l5, "flac -5 (the tool's default preset) on the music-like signal"),
                                     ("level0", l0, "flac -0 (fixed predictors only, 1152-sample blocks, no mid/side) on the music-like signal"),
                                     ("hires", hr, "flac -8 on 96 kHz / 24-bit stereo (BASELINE.json config 4: the wide-sample residual path), 4096-sample blocks")):
                    extras[key] = {"what": what, "value": round(r["value"], 3), "unit": "Msamples/s", "ms_per_step": round(r["ms_per_step"], 4), "steps": r["steps"], "frames_per_step": r.get("nframes"),
                                   "compressed_bytes_per_sample": round(r["out_bps"], 4), "kernel_ms": {k: round(v, 4) for k, v in r["kernel_ms"].items()},
                                   "clock": r.get("clock"), "roofline": r["roofline"], "roofline_valu": r.get("roofline_valu"), "verified_frames": r.get("verified", {}).get("frames_compared_with_oracle"),
                                   "verified_ok": r.get("verified", {}).get("ok")}
        except Exception as e:
            args.hires = False
            RATE, BPS = 44100, 16
            extras["side_measurements_error"] = "%s: %s" % (type(e).__name__, e)

    if rank == 0:
        line = compose(extras, side)
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(LEVEL, search)
                line["cpu_baseline"] = cb
                line["speedup_vs_cpu_1thread"] = round(m["value"] / cb["value"], 2)
                if "all_cores" in cb:
                    line["speedup_vs_cpu_allcores"] = round(m["value"] / cb["all_cores"]["value"], 2)
                    if cb["library_thread_pool"]["value"]:
                        line["speedup_vs_cpu_library_pool"] = round(m["value"] / cb["library_thread_pool"]["value"], 2)
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        emit(line)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
