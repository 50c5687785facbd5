"""gloo runs (world sizes 2, 4 and 8) of the multi-GPU path's only exchange step: the ordered variable-length gather of encoded
shards to rank 0 (flac_amd/dist.py) -- the one-shot form, the form that receives into a preallocated buffer the
destination encoded into, and the windowed steady-state pipeline -- plus the frame-range sharding rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flac_amd.dist import GatherPipeline, HostShmGather, HostShmPipeline, ordered_gather, shard_range


def test_shard_range_partitions_exactly():
    for nframes in (0, 1, 7, 8, 9, 1000, 387598):
        for world in (1, 2, 4, 8):
            spans = [shard_range(nframes, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nframes
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _shard(rank):
    rng = np.random.default_rng(100 + rank)
    nfr = 0 if rank == 2 else 5 + 3 * rank                                # ragged; rank 2 (world >= 4) has no frame at all
    fb = rng.integers(14, 9000, nfr).astype(np.int32)
    nbytes = int(fb.sum())
    return fb, nbytes, rng.integers(0, 256, nbytes, dtype=np.uint8)


def _step_shard(rank, k, nframes, cap):
    """what rank `rank` 'encodes' in step k of the pipeline test"""
    rng = np.random.default_rng(1000 * k + rank)
    fb = rng.integers(14, cap // nframes, nframes).astype(np.int32)
    if rank == 3 and k % 2 == 1:
        fb[:] = 0                                                           # a rank that produced nothing in this step
    nbytes = int(fb.sum())
    return fb, nbytes, rng.integers(0, 256, nbytes, dtype=np.uint8)


def _worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode in ("oneshot", "prealloc"):
            fb, nbytes, data = _shard(rank)
            if mode == "oneshot":
                payload = np.zeros(nbytes + 77, dtype=np.uint8)       # capacity larger than the used bytes
                payload[:nbytes] = data
                stream, allfb = ordered_gather(torch.from_numpy(payload), nbytes, torch.from_numpy(fb), dst=0)
            else:
                # the destination "encoded" straight into the head of the receive buffer; the byte count is a tensor
                out = torch.zeros(2000000, dtype=torch.uint8)
                outfb = torch.zeros(256, dtype=torch.int64)
                payload = out[:nbytes + 5] if rank == 0 else torch.zeros(nbytes + 5, dtype=torch.uint8)
                payload[:nbytes] = torch.from_numpy(data)
                stream, allfb = ordered_gather(payload, torch.tensor([nbytes]), torch.from_numpy(fb), dst=0, out=out if rank == 0 else None,
                                               out_frame_bytes=outfb if rank == 0 else None)
                if rank == 0:
                    assert stream.data_ptr() == out.data_ptr()
            if rank == 0:
                q.put((stream.numpy().tobytes(), allfb.numpy().tolist()))
            else:
                assert stream is None and allfb is None
        elif mode == "hostshm":
            fb, nbytes, data = _shard(rank)
            hs = HostShmGather(1 << 20)
            off, sizes = hs.gather(torch.from_numpy(np.concatenate([data, np.zeros(9, np.uint8)])), nbytes)
            if rank == 0:
                q.put((hs.host[:sum(sizes)].numpy().tobytes(), sizes, oct(os.stat(hs.path).st_mode & 0o777)))
            hs.close()
        elif mode[0] == "hostpipe":
            _, window, steps = mode
            nframes, cap = 6, 60000
            gp = HostShmPipeline(cap, nframes, "cpu", window=window)
            for k in range(steps):
                gp.wait_slot_free(k)
                out, fbt, total = gp.slot(k)
                fb, nbytes, data = _step_shard(rank, k, nframes, cap)
                out[:nbytes] = torch.from_numpy(data)
                fbt.copy_(torch.from_numpy(fb))
                total[0] = nbytes
                gp.step_done(k)
            gp.flush()
            dist.barrier()
            # the last two windows are still in their slots: every rank sees the same bytes
            got = {}
            for kk in range(max(0, ((steps - 1) // window - 1) * window), steps):
                sv, sz, _ = gp.gathered(kk)
                got[kk] = (sv.numpy().tobytes(), sz)
            if rank == world - 1:
                q.put(got)
            assert gp.host_syncs == (steps + window - 1) // window
            assert len(gp.window_stats()) == gp.gathers
            gp.close()
        else:
            window, steps = mode
            nframes, cap = 6, 60000
            gp = GatherPipeline(cap, nframes, "cpu", window=window)
            got = {}

            for k in range(steps):
                gp.wait_slot_free(k)
                out, fbt, total = gp.slot(k)
                fb, nbytes, data = _step_shard(rank, k, nframes, cap)
                out[:nbytes] = torch.from_numpy(data)
                fbt.copy_(torch.from_numpy(fb))
                total[0] = nbytes
                before = gp.gathers
                gp.step_done(k)
                if rank == 0 and gp.gathers != before:
                    # the window gathered just now is the one before the window that just closed
                    k0 = (k + 1 - 2 * window)
                    for kk in range(k0, k0 + window):
                        s, sz, fbs = gp.gathered(kk)
                        got[kk] = (s.numpy().tobytes(), sz, fbs.numpy().tolist())
            done_before = set(got)
            gp.flush()
            if rank == 0:
                for kk in range(steps):
                    if kk not in done_before:
                        s, sz, fbs = gp.gathered(kk)
                        got[kk] = (s.numpy().tobytes(), sz, fbs.numpy().tolist())
                assert gp.host_syncs == (steps + window - 1) // window         # one host read of sizes per window
                q.put(got)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(mode, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("mode", ["oneshot", "prealloc"])
def test_ordered_gather_gloo_ws2(mode):
    world = 2
    got_stream, got_fb = _run(mode, world)
    want_stream, want_fb = b"", []
    for rank in range(world):
        fb, nbytes, data = _shard(rank)
        want_stream += data.tobytes()
        want_fb += fb.tolist()
    assert got_fb == want_fb
    assert got_stream == want_stream


@pytest.mark.parametrize("window,steps", [(1, 5), (3, 7), (4, 8)])
def test_gather_pipeline_gloo_ws2(window, steps):
    world = 2
    got = _run((window, steps), world)
    assert sorted(got) == list(range(steps))
    for k in range(steps):
        stream, sizes, fbs = got[k]
        want = b""
        for rank in range(world):
            fb, nbytes, data = _step_shard(rank, k, 6, 60000)
            want += data.tobytes()
            assert sizes[rank] == nbytes
            assert fbs[rank] == fb.tolist()
        assert stream == want


# ---- more than one predecessor per rank: the cumulative offsets of ranks >= 2 ----------------------------------------------------
@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("mode", ["oneshot", "prealloc"])
def test_ordered_gather_gloo_ragged(world, mode):
    got_stream, got_fb = _run(mode, world)
    want_stream, want_fb = b"", []
    for rank in range(world):
        fb, nbytes, data = _shard(rank)
        want_stream += data.tobytes()
        want_fb += fb.tolist()
    assert got_fb == want_fb
    assert got_stream == want_stream


def _check_pipeline(got, world, steps):
    assert sorted(got) == list(range(steps))
    for k in range(steps):
        stream, sizes, fbs = got[k]
        want = b""
        for rank in range(world):
            fb, nbytes, data = _step_shard(rank, k, 6, 60000)
            want += data.tobytes()
            assert sizes[rank] == nbytes
            assert fbs[rank] == fb.tolist()
        assert stream == want


@pytest.mark.parametrize("world,window,steps", [(4, 3, 7), (4, 4, 9), (8, 2, 5), (8, 4, 6)])
def test_gather_pipeline_gloo_many_ranks(world, window, steps):
    """window counts that do not divide the step count, a rank (3) that has nothing to send every other step"""
    _check_pipeline(_run((window, steps), world), world, steps)


@pytest.mark.parametrize("world", [2, 4])
def test_host_shm_gather_gloo(world):
    stream, sizes, mode = _run("hostshm", world)
    want = b"".join(_shard(r)[2].tobytes() for r in range(world))
    assert stream == want and sizes == [_shard(r)[1] for r in range(world)]
    assert mode == "0o600"                                   # nobody else may read the stream


@pytest.mark.parametrize("world,window,steps", [(2, 2, 5), (4, 3, 7), (8, 2, 4)])
def test_host_shm_pipeline_gloo(world, window, steps):
    got = _run(("hostpipe", window, steps), world)
    assert max(got) == steps - 1
    for k, (stream, sizes) in got.items():
        want = b""
        for rank in range(world):
            fb, nbytes, data = _step_shard(rank, k, 6, 60000)
            want += data.tobytes()
            assert sizes[rank] == nbytes
        assert stream == want
