#!/usr/bin/env python
"""The logging collective of SURVEY.md 8e on the real backend: run under torch.distributed.run (backend "nccl" = RCCL), one rank
per GPU.  Every rank steps its shard, snapshots its [N, 3] statistics block and all-gathers it on a side stream
(ShardedWorld.log_episode_stats); the gathered block must equal the concatenation of every rank's own
hh_episode_stats_packed block (exchanged once more with a plain all_gather on the default stream as the check).
Prints RCCL_CHECK_OK <ranks> <rows> on rank 0.   tests/test_gpu_sharding_rccl.py runs it with one rank on the 1-GPU box;
`python tools/rccl_check.py --launch 8` is the 8-GPU form: it starts the job with NCCL_DEBUG=INFO and also fails when RCCL's own
log reports fewer ranks than were launched."""
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
    ev = sw.evidence()
    assert ev["ranks_seen"] == world and ev["gathered_rows"] == world * N and ev["first_global_arena_of_each_block"] == [r * N for r in range(world)], ev
    dist.barrier()
    if rank == 0:
        print(f"RCCL_CHECK_OK {world} {got.shape[0]} finished={finished}", flush=True)
    dist.destroy_process_group()


def launch(n):
    """`python tools/rccl_check.py --launch N`: start the N-rank job itself with NCCL_DEBUG=INFO (one log file per rank) and FAIL unless RCCL's
    own log says what the script printed: every rank's communicator finished its init with `nranks N`, and N distinct ranks did."""
    import re
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    logdir = tempfile.mkdtemp(prefix="rccl_check_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT",
               NCCL_DEBUG_FILE=os.path.join(logdir, "rccl.%h.%p.log"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    sys.stdout.write(p.stdout)
    sys.stderr.write(p.stderr[-4000:])
    if p.returncode != 0 or f"RCCL_CHECK_OK {n} " not in p.stdout:
        sys.exit(f"rccl_check: the {n}-rank job failed (exit {p.returncode})")
    ranks, nranks = set(), set()
    for fn in os.listdir(logdir):
        with open(os.path.join(logdir, fn), errors="replace") as f:
            for ln in f:
                m = re.search(r"rank (\d+) nranks (\d+).*Init COMPLETE", ln)
                if m:
                    ranks.add(int(m.group(1)))
                    nranks.add(int(m.group(2)))
    # the job may create more than one communicator (the default group and device-bound ones); the one that spans the job has nranks == n
    if n not in nranks or not set(range(n)) <= ranks:
        sys.exit(f"rccl_check: RCCL's log reports ranks {sorted(ranks)} and communicator sizes {sorted(nranks)}; {n} ranks were launched ({logdir})")
    print(f"RCCL_LOG_OK ranks={sorted(ranks)} nranks={sorted(nranks)}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--launch":
        launch(int(sys.argv[2]))
    else:
        main()
