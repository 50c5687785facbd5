#!/usr/bin/env python3
"""bench.py -- encode throughput of the MI355X FLAC frame engine at -8, 44.1 kHz/16-bit stereo.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (flacgpu_encode_batch_device: analyze -> pack -> scan/compact)
over one batch of synthetic PCM that is already resident in HBM.  With N > 1 every rank encodes its
own contiguous frame range (weak scaling, no data-path collective) and the step ends with the ordered
RCCL gather of the variable-length bitstream to rank 0 (flac_amd/dist.py).

Prints ONE JSON line (rank 0).  `value` = inter-channel samples encoded per second by the whole job,
in M samples/s; `roofline` prices the dominant kernel of the step (by HIP-event time) against HBM bandwidth with the
ALGORITHMIC bytes of SURVEY.md 8d (4*C bytes of PCM in + compressed bytes out per inter-channel
sample); `cpu_baseline` is the unmodified reference libFLAC (oracle/_ref, AVX2+FMA dispatch) timed on
this box's host cores on a bounded sample of the same signal.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LEVEL = 8
RATE, BPS, CH = 44100, 16, 2
BLOCK = 4096
FRAMES_PER_GPU = 16384         # 67.1 M inter-channel samples = 25 min of audio per GPU per step (2.7 GB of HBM in all)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
KERNEL_NAMES = {"prep": "prep2_kernel", "autoc": "autoc2_kernel", "model": "model_kernel", "eval": "eval_kernel",
                "pack": "pack2_kernel", "scan_compact": "scan_kernel+compact_kernel"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from the committed rocprofv3 PMC passes


def synth_pcm(nframes, seed):
    """music-like 16-bit stereo (tests/signals.py: tones + coloured noise, channels correlated)"""
    import signals
    base_frames = min(nframes, 512)
    base = signals.music(base_frames * BLOCK, CH, BPS, seed=seed, rate=RATE)
    reps = (nframes + base_frames - 1) // base_frames
    if reps > 1:
        # repeat the 48 s clip with a per-repeat gain/offset so frames are not byte-identical
        parts = []
        for r in range(reps):
            g = 1.0 - 0.07 * (r % 8)
            parts.append(np.clip(np.rint(base * g) + (r % 5) - 2, -(1 << (BPS - 1)), (1 << (BPS - 1)) - 1).astype(np.int32))
        base = np.concatenate(parts, axis=0)
    return np.ascontiguousarray(base[: nframes * BLOCK])


def cpu_baseline(sample_pcm, search=None):
    """Reference libFLAC on the host cores: single thread, then many independent encoders."""
    from oracle import pyoracle as po
    from concurrent.futures import ThreadPoolExecutor
    n = sample_pcm.shape[0]
    if po.have_ref():
        kind = "reference"

        def run():
            return po.ref_encode(sample_pcm, BPS, RATE, LEVEL, want_bytes=False, **(search or {}))["seconds"]
    else:
        kind = "port"

        def run():
            t0 = time.perf_counter()
            po.oracle_encode(sample_pcm, BPS, RATE, LEVEL, **(search or {}))
            return time.perf_counter() - t0
    best = min(run() for _ in range(3))
    single = n / best / 1e6
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:     # ctypes releases the GIL: independent encoders in parallel
        list(ex.map(lambda _: run(), range(threads * 2)))
    multi = threads * 2 * n / (time.perf_counter() - t0) / 1e6
    # the library's own frame-parallel thread pool (flac -j N, stream_encoder.c:2151) on one longer stream
    pool = None
    if kind == "reference":
        try:
            long_pcm = np.concatenate([sample_pcm] * 8, axis=0)
            nthr = min(threads, 32)
            sec = min(po.ref_encode(long_pcm, BPS, RATE, LEVEL, want_bytes=False, num_threads=nthr, **(search or {}))["seconds"] for _ in range(2))
            pool = {"value": round(long_pcm.shape[0] / sec / 1e6, 3), "threads": nthr, "how": "one stream, FLAC__stream_encoder_set_num_threads"}
        except Exception as e:           # a reference build without threads
            pool = {"error": str(e)}
    return {
        "value": round(single, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "sample": "%d inter-channel samples (%.0f s) of the bench signal, flac -%d, best of 3, in-memory" % (n, n / RATE, LEVEL),
        "multi": {"value": round(multi, 3), "cores": threads, "how": "independent single-thread encoders, 2 clips each"},
        "library_thread_pool": pool,
        "host_cpus": cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hires", action="store_true", help="96 kHz / 24-bit stereo (BASELINE.json config 4): a side measurement")
    ap.add_argument("--exhaustive", action="store_true", help="flac -8e: not the headline workload, a side measurement")
    ap.add_argument("--prec-search", action="store_true", help="flac -8p")
    ap.add_argument("--level", type=int, default=8, help="compression preset -0..-8 (the metric is quoted at -8: other levels are side measurements; "
                    "-0..-2 use the preset's 1152-sample blocks)")
    ap.add_argument("--force-dist", action="store_true", help="development aid: run the multi-rank pipeline (process group, "
                    "overlapped ordered gather) even with one rank")
    args = ap.parse_args()
    global RATE, BPS, LEVEL, BLOCK
    if args.hires:
        RATE, BPS = 96000, 24
    if args.level != 8:
        LEVEL = args.level
        BLOCK = 1152 if LEVEL < 3 else 4096

    # stdout must carry exactly one JSON line: libraries underneath (RCCL prints a version banner from C) get stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import flac_amd
    from flac_amd.dist import ordered_gather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nframes = args.frames
    search = dict(exhaustive=int(args.exhaustive), prec_search=int(args.prec_search))
    settings = flac_amd.make_settings(CH, BPS, RATE, LEVEL, **search)
    eng = flac_amd.FrameEngine(settings, device=local_rank, max_batch_frames=nframes)

    # this rank's shard of the corpus: frames [rank*nframes, (rank+1)*nframes)
    pcm_h = synth_pcm(nframes, seed=1234 + rank)
    d_pcm = torch.from_numpy(pcm_h).to(dev)
    cap = eng.max_output_bytes(nframes)
    first_frame = rank * nframes
    phase_ms = []

    if not multi:
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_fb = torch.empty(nframes, dtype=torch.int32, device=dev)
        d_total = torch.zeros(1, dtype=torch.int64, device=dev)
        enc_stream = torch.cuda.Stream()

        def run(nsteps):
            for _ in range(nsteps):
                eng.encode_device(d_pcm.data_ptr(), nframes, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(),
                                  first_frame_number=first_frame, stream=enc_stream.cuda_stream)
    else:
        # Two output buffers: the ordered RCCL gather of step k (its own stream) overlaps the encode of step k+1.
        # Only the gather's stream is ever synchronised with the host (for the byte count it has to send).
        enc_stream, comm_stream = torch.cuda.Stream(), torch.cuda.Stream()
        bufs = [{"out": torch.empty(cap, dtype=torch.uint8, device=dev), "fb": torch.empty(nframes, dtype=torch.int32, device=dev),
                 "total": torch.zeros(1, dtype=torch.int64, device=dev), "enc_done": torch.cuda.Event(), "free": torch.cuda.Event(),
                 "used": False} for _ in range(2)]
        d_total = bufs[0]["total"]

        def launch_encode(k):
            b = bufs[k % 2]
            if b["used"]:
                enc_stream.wait_event(b["free"])        # the gather that read this buffer is done
            eng.encode_device(d_pcm.data_ptr(), nframes, b["out"].data_ptr(), cap, b["fb"].data_ptr(), b["total"].data_ptr(),
                              first_frame_number=first_frame, stream=enc_stream.cuda_stream)
            b["enc_done"].record(enc_stream)
            b["used"] = True

        def gather(k):
            b = bufs[k % 2]
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(b["enc_done"])
                nbytes = int(b["total"].item())          # host waits for encode k only; encode k+1 is already queued
                ordered_gather(b["out"], nbytes, b["fb"], dst=0)
                b["free"].record(comm_stream)

        def run(nsteps):
            launch_encode(0)
            for k in range(1, nsteps):
                launch_encode(k)
                gather(k - 1)
            gather(nsteps - 1)

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    # per-kernel durations of the timed steps: HIP events the engine recorded on its stream around every launch (it keeps
    # the sets of its last 64 batches, so nothing had to sync inside the timed region)
    for back in range(min(args.steps, 64)):
        phase_ms.append(eng.last_phase_ms(back))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_bytes = int(d_total.item())
    samples_per_step = nframes * BLOCK
    ms_per_step = elapsed / args.steps * 1e3
    value = world * samples_per_step * args.steps / elapsed / 1e6

    if rank == 0:
        out_bps = total_bytes / samples_per_step                 # compressed bytes per inter-channel sample
        alg_bytes = samples_per_step * (4 * CH + out_bps)         # PCM read once + frames written once
        kms = {k: float(np.mean([ph[k] for ph in phase_ms])) for k in phase_ms[0]}
        dom = max(kms, key=kms.get)                               # the dominant kernel of the step
        achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9
        traffic = None
        try:
            with open(PMC_FILE) as fh:
                pmc = json.load(fh)
            per_frame = pmc["kernels"][KERNEL_NAMES[dom]]["hbm_bytes_per_frame"]
            traffic = int(per_frame * nframes)
        except Exception:
            pass
        line = {
            "metric": ("encode Msamples/s at -8, 44.1k/16-bit stereo; bit-exact vs libFLAC" if not args.hires else "encode Msamples/s at -8, 96k/24-bit stereo (side measurement)")
                      if LEVEL == 8 else "encode Msamples/s at -%d (side measurement; the metric is quoted at -8)" % LEVEL,
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": "flac -%d%s%s (%s) on %s stereo, "
                                   "%d frames x %d samples per GPU per step, music-like synthetic PCM resident in HBM" % (LEVEL, "e" if args.exhaustive else "", "p" if args.prec_search else "",
                                   "max LPC order 12, subdivide_tukey(3), mid/side, partition order <= 6" if LEVEL == 8 else "the preset's settings, stream_encoder.c:117-140",
                                   "96k/24-bit" if args.hires else "44.1k/16-bit", nframes, BLOCK),
                       "frames_per_gpu_per_step": nframes, "blocksize": BLOCK, "channels": CH, "bits_per_sample": BPS,
                       "samples_are": "inter-channel (x2 for channel-samples)", "parallelism": "frame-shard x%d + ordered RCCL gather of every step's frames to rank 0, overlapped with the next step's encode" % world,
                       "compressed_bytes_per_sample": round(out_bps, 4)},
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "-8 is VALU bound (~1e3 integer+fp64 ops per sample; the dominant kernels issue VALU work 77-83% of their cycles, "
                                 "profiles/*pmc*); the HBM fraction is reported because the north star asks for it; traffic = FETCH_SIZE+WRITE_SIZE "
                                 "of the committed PMC pass scaled to this batch"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pcm_h[: (512 if not (args.exhaustive or args.prec_search) else 64) * BLOCK], search)
            line["speedup_vs_cpu_1thread"] = round(value / line["cpu_baseline"]["value"], 2)
            line["speedup_vs_cpu_multi"] = round(value / line["cpu_baseline"]["multi"]["value"], 2)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    eng.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
