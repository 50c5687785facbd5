"""CPU: the check bench.py runs on a multi-rank line (bench.verify_ranks) -- every rank's frames of the gathered step, not only rank
0's (ranks 1..7 reached the JSON unchecked).  Four gloo ranks 'encode' their bench signal (bench.rank_pcm, seed 1234 +
rank) with the ORACLE, frame numbers rank * frames + f, into the slots of the real GatherPipeline; rank 0 checks what it gathered.
Then the same with damage: a flipped bit (CRC-16), two ranks' segments in the wrong order, a rank that numbered its frames wrongly,
frame lengths that do not add up -- each must be pinned on the right rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NFRAMES, LEVEL, BLOCK = 6, 0, 1152


def _encode(rank, first_frame):
    import bench
    from oracle import pyoracle as po
    pcm = bench.rank_pcm(rank, NFRAMES, BLOCK, "music", False)
    o = po.oracle_encode(pcm, 16, 44100, LEVEL, first_frame=first_frame)
    return np.frombuffer(o["data"], dtype=np.uint8), np.asarray(o["frame_bytes"], dtype=np.int32)


def _worker(rank, world, port, q, damage):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from flac_amd.dist import GatherPipeline
        cap = 4 * NFRAMES * BLOCK * 2 + 4096
        gp = GatherPipeline(cap, NFRAMES, "cpu", window=2)
        steps = 3
        for k in range(steps):
            gp.wait_slot_free(k)
            out, fbt, total = gp.slot(k)
            first = rank * NFRAMES + (1 if damage == "numbering" and rank == 3 else 0)
            data, fb = _encode(rank, first)
            out[:data.size] = torch.from_numpy(data.copy())
            fbt.copy_(torch.from_numpy(fb))
            total[0] = int(data.size)
            gp.step_done(k)
        gp.flush()
        if rank == 0:
            stream, sizes, fbs = gp.gathered(steps - 1)
            stream = stream.numpy().copy()
            fbs = fbs.numpy().copy()
            if damage == "bitflip":
                stream[sum(sizes[:2]) + sizes[2] // 2] ^= 0x10          # somewhere inside rank 2's segment
            elif damage == "order":
                a, b = sum(sizes[:1]), sum(sizes[:2])
                seg1, seg2 = stream[a:b].copy(), stream[b:b + sizes[2]].copy()
                stream[a:a + sizes[2]] = seg2
                stream[a + sizes[2]:a + sizes[2] + sizes[1]] = seg1
                sizes = [sizes[0], sizes[2], sizes[1], sizes[3]]
                fbs[[1, 2]] = fbs[[2, 1]]
            elif damage == "lengths":
                fbs[1, 0] += 1
            q.put(bench.verify_ranks(stream, sizes, fbs, NFRAMES, LEVEL, BLOCK, nsample=4))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(damage, world=4):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, damage)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return res


def test_every_rank_of_a_gathered_step_is_checked():
    v = _run(None)
    assert v["ranks_checked"] == 4 and v["ok"] and v["ranks_failing"] == []
    assert v["crc16_frames_checked"] == 4 * NFRAMES and v["frames_compared_with_oracle"] >= 4 * 4


@pytest.mark.parametrize("damage,bad", [("bitflip", [2]), ("order", [1, 2]), ("numbering", [3]), ("lengths", [1])])
def test_damage_is_pinned_on_its_rank(damage, bad):
    v = _run(damage)
    assert v["ranks_checked"] == 4 and not v["ok"]
    assert v["ranks_failing"] == bad
