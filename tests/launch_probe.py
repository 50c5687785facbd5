"""A job that starts its own ranks the way bench.py and flac_amd.corpus do (flac_amd.dist.ensure_ranks), on the gloo backend so
that it runs without a GPU: rank 0 prints ONE JSON line with what the group saw.  Used by tests/test_launch_cpu.py."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--need-devices", action="store_true")
    args = ap.parse_args()
    from flac_amd.dist import ensure_ranks, check_world, shard_range
    rank, local_rank, world = ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], need_devices=args.need_devices)
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        check_world(args.gpus)
        t = torch.tensor([rank, local_rank, 1], dtype=torch.int64)
        table = [torch.empty(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(table, t)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "world_size_seen": dist.get_world_size(), "ranks": [int(x[0]) for x in table], "local_ranks": [int(x[1]) for x in table],
                              "shards": [list(shard_range(1000, world, r)) for r in range(world)], "self_launched": "TORCHELASTIC_RUN_ID" in os.environ}))
        dist.barrier()
        dist.destroy_process_group()
    else:
        print(json.dumps({"n_gpus": 1, "world_size_seen": 1, "ranks": [0], "local_ranks": [0], "shards": [[0, 1000]], "self_launched": False}))


if __name__ == "__main__":
    main()
