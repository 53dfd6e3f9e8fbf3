#!/usr/bin/env python
"""The logging collective of SURVEY.md 8e on the real backend: run under torch.distributed.run (backend "nccl" = RCCL), one rank
per GPU.  Every rank steps its shard, snapshots its [N, 3] statistics block and all-gathers it on a side stream
(ShardedWorld.log_episode_stats); the gathered block must equal the concatenation of every rank's own
hh_episode_stats_packed block (exchanged once more with a plain all_gather on the default stream as the check).
Prints RCCL_CHECK_OK <ranks> <rows> on rank 0.   tests/test_gpu_sharding_rccl.py runs it with one rank on the 1-GPU box;
`python -m torch.distributed.run --nproc-per-node 8 tools/rccl_check.py` is the 8-GPU form."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from hhmarl_2d_amd.sharding import ShardedWorld
    N, T = 512, 60
    sw = ShardedWorld(dict(n_arenas=N, level=3, seed=11, auto_reset=True, horizon=40), rank=rank, world_size=world, device=local)
    w = sw.world
    w.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(5 + rank)
    act = (torch.rand((T, N, 2, 4), device=dev, generator=g) * torch.tensor([13, 9, 2, 2], device=dev)).to(torch.int8).contiguous()
    side = torch.cuda.Stream()
    for k in range(4):
        w.rollout(act)
        sw.log_episode_stats(side)          # snapshot on the stepping stream, all-gather on the side stream
    got = sw.wait_stats().clone()           # ordered after the side-stream gather
    mine = w.episode_stats_packed().clone() # nothing stepped since the last snapshot: the same block
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    want = torch.cat(parts, dim=0)
    torch.cuda.synchronize()
    assert got.shape == (world * N, 3), got.shape
    assert torch.equal(got, want), "side-stream all-gather differs from the ranks' own statistics blocks"
    assert torch.equal(got[rank * N:(rank + 1) * N], mine)
    finished = int((got[:, 2] != 2).sum())
    assert finished > 0, "no episode finished: the block would be trivially equal"
    dist.barrier()
    if rank == 0:
        print(f"RCCL_CHECK_OK {world} {got.shape[0]} finished={finished}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
