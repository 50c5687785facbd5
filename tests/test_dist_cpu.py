"""world_size-2 gloo run of the multi-GPU path's only exchange step: the ordered variable-length
gather of encoded shards to rank 0 (flac_amd/dist.py), plus the frame-range sharding rule."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flac_amd.dist import ordered_gather, shard_range


def test_shard_range_partitions_exactly():
    for nframes in (0, 1, 7, 8, 9, 1000, 387598):
        for world in (1, 2, 4, 8):
            spans = [shard_range(nframes, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nframes
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        nfr = 5 + 3 * rank
        fb = rng.integers(14, 9000, nfr).astype(np.int32)
        nbytes = int(fb.sum())
        payload = np.zeros(nbytes + 77, dtype=np.uint8)       # capacity larger than the used bytes
        payload[:nbytes] = rng.integers(0, 256, nbytes, dtype=np.uint8)
        stream, allfb = ordered_gather(torch.from_numpy(payload), nbytes, torch.from_numpy(fb), dst=0)
        if rank == 0:
            q.put((stream.numpy().tobytes(), allfb.numpy().tolist()))
        else:
            assert stream is None and allfb is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ordered_gather_gloo_ws2():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got_stream, got_fb = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_stream, want_fb = b"", []
    for rank in range(world):
        rng = np.random.default_rng(100 + rank)
        nfr = 5 + 3 * rank
        fb = rng.integers(14, 9000, nfr).astype(np.int32)
        nbytes = int(fb.sum())
        want_stream += rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
        want_fb += fb.tolist()
    assert got_fb == want_fb
    assert got_stream == want_stream
