"""flac_amd.dist -- multi-GPU sharding of a frame-parallel encode job.

Frames are independent, so a corpus shards by contiguous frame ranges with NO data-path collective:
rank r encodes frames [r*F/W, (r+1)*F/W) with the frame numbers fixed up front.  The only exchange is
the final ORDERED gather of the variable-length bitstream to rank 0:
    all_gather(byte totals, frame counts) -> exclusive scan -> point-to-point payload sends
i.e. a Gatherv built from RCCL send/recv over xGMI (works identically over gloo on CPU, which is how
tests/test_dist_cpu.py covers it).
"""
import torch
import torch.distributed as dist


def shard_range(nframes, world, rank):
    """Contiguous frame range [lo, hi) of `rank`; earlier ranks take the remainder."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def ordered_gather(payload, nbytes, frame_bytes, group=None, dst=0):
    """Gather variable-length encoded shards to `dst` in rank order.

    payload     uint8 tensor (device or CPU) whose first `nbytes` bytes are this rank's frames
    frame_bytes uint32/int32 tensor [nframes_local] with the length of each local frame
    Returns on dst: (stream uint8 tensor [total], all_frame_bytes int64 tensor [total frames]);
    on other ranks: (None, None).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    nfr = int(frame_bytes.numel())
    meta = torch.tensor([int(nbytes), nfr], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    sizes = [int(m[0].item()) for m in metas]
    counts = [int(m[1].item()) for m in metas]
    fb = frame_bytes.to(torch.int64)
    if rank == dst:
        total, totalf = sum(sizes), sum(counts)
        stream = torch.empty(total, dtype=torch.uint8, device=dev)
        allfb = torch.empty(totalf, dtype=torch.int64, device=dev)
        ops, off, foff = [], 0, 0
        for r in range(world):
            if r == rank:
                stream[off:off + sizes[r]].copy_(payload[:sizes[r]])
                allfb[foff:foff + counts[r]].copy_(fb)
            else:
                if sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, stream[off:off + sizes[r]], r, group))
                if counts[r]:
                    ops.append(dist.P2POp(dist.irecv, allfb[foff:foff + counts[r]], r, group))
            off += sizes[r]
            foff += counts[r]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return stream, allfb
    ops = []
    if nbytes:
        ops.append(dist.P2POp(dist.isend, payload[:nbytes].contiguous(), dst, group))
    if nfr:
        ops.append(dist.P2POp(dist.isend, fb.contiguous(), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None, None
