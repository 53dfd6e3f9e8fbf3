"""Multi-GPU path on CPU: arena sharding is bit-invariant (a 2-rank sharded run equals one big
world) and the logging all-gather returns the per-arena statistics in global arena order.
Runs world_size-2 `gloo` process groups; the per-rank world is backed by the CPU oracle through
ShardedWorld's world_factory hook (tests may use the oracle; the product path never does)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import random_actions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleBackedWorld:
    """adapter: the World call surface on top of oracle_lib.OracleWorld with torch tensors"""

    def __init__(self, kw):
        import oracle_lib as O
        self.o = O.OracleWorld(O.make_config(**kw))
        self.N, self.n_ctrl = self.o.N, self.o.n_ctrl

    def reset(self):
        return torch.from_numpy(self.o.reset())

    def rollout(self, act):
        return [torch.from_numpy(x) for x in self.o.rollout(act.numpy())]

    def episode_stats(self):
        return [torch.from_numpy(x) for x in self.o.episode_stats()]


def _worker(rank, world_size, port, n_per_rank, T, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from hhmarl_2d_amd.sharding import ShardedWorld, summarize
    kw = dict(n_arenas=n_per_rank, level=3, seed=42, auto_reset=True, horizon=40)
    sw = ShardedWorld(kw, rank=rank, world_size=world_size, world_factory=_OracleBackedWorld)
    sw.world.reset()
    act_all = random_actions(np.random.default_rng(0), (T, n_per_rank * world_size), 2)
    act = torch.from_numpy(np.ascontiguousarray(act_all[:, rank * n_per_rank:(rank + 1) * n_per_rank]))
    obs, rew, val, done = sw.world.rollout(act)
    stats = sw.log_episode_stats()
    assert stats.shape == (n_per_rank * world_size, 3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), obs=obs.numpy(), rew=rew.numpy(), done=done.numpy(), stats=stats.numpy())
    s = summarize(stats)
    assert s["episodes"] > 0 and s["agents_win"] + s["opps_win"] + s["draw"] == s["episodes"]
    # what a multi-rank bench line quotes: the collective itself saw every rank, blocks in rank order tiling the global arena range
    ev = sw.evidence()
    assert ev["ranks_seen"] == world_size and ev["gathered_rows"] == n_per_rank * world_size and ev["backend"] == "gloo"
    assert ev["first_global_arena_of_each_block"] == [r * n_per_rank for r in range(world_size)]
    # a world sharded for MORE ranks than the job has must not pass for a global block
    from hhmarl_2d_amd.sharding import gather_stats
    with pytest.raises(RuntimeError, match="ranks"):
        gather_stats(stats[:n_per_rank].contiguous(), world_size + 1)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_one_world(tmp_path, oracle):
    n_per_rank, T, ws = 24, 60, 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(ws, port, n_per_rank, T, str(tmp_path)), nprocs=ws, join=True)
    # single world with all arenas
    o = oracle.OracleWorld(oracle.make_config(n_arenas=n_per_rank * ws, level=3, seed=42, auto_reset=True, horizon=40))
    o.reset()
    act_all = random_actions(np.random.default_rng(0), (T, n_per_rank * ws), 2)
    obs, rew, val, done = o.rollout(act_all)
    ret, ln, oc = o.episode_stats()
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(ws)]
    assert np.array_equal(np.concatenate([x["obs"] for x in r], axis=1), obs)
    assert np.array_equal(np.concatenate([x["rew"] for x in r], axis=1), rew)
    assert np.array_equal(np.concatenate([x["done"] for x in r], axis=1), done)
    want = np.stack([ret, ln.astype(np.float32), oc.astype(np.float32)], axis=1)
    for x in r:  # every rank holds the full gathered table, in global arena order
        assert np.array_equal(x["stats"], want)


def _hier_run(o, cmd, tape):
    """commander steps of an oracle HighLevelEnv world with taped pilot actions -> stacked (obs, reward, done)"""
    outs = []
    for c in cmd:
        o.hl_begin(c)
        for k in range(16):
            o.hl_agents_act(tape[k])
            if o.hl_tick(tape[k]) == 0:
                pass   # fixed 16 sub-steps like the graph-captured macro step: finished arenas idle
        ob, rw, vl, dn = o.hl_end()
        outs.append((ob, rw, dn))
    return [np.stack([x[i] for x in outs]) for i in range(3)]


def _worker_hier(rank, world_size, port, n_per_rank, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from hhmarl_2d_amd.sharding import ShardedWorld, summarize
    kw = dict(n_arenas=n_per_rank, env_kind=1, seed=42, auto_reset=True, horizon=48)   # configs[4] shape: 3-vs-3 HighLevelEnv shards
    sw = ShardedWorld(kw, rank=rank, world_size=world_size, world_factory=_OracleBackedWorld)
    sw.world.reset()
    rng = np.random.default_rng(0)
    cmd_all = rng.integers(0, 3, (steps, n_per_rank * world_size, 3)).astype(np.int8)
    tape_all = random_actions(rng, (16, n_per_rank * world_size), 6)
    lo, hi = rank * n_per_rank, (rank + 1) * n_per_rank
    obs, rew, done = _hier_run(sw.world.o, np.ascontiguousarray(cmd_all[:, lo:hi]), np.ascontiguousarray(tape_all[:, lo:hi]))
    stats = sw.log_episode_stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), obs=obs, rew=rew, done=done, stats=stats.numpy())
    assert summarize(stats)["episodes"] > 0
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_hier_equals_one_world(tmp_path, oracle):
    """BASELINE configs[4]'s shape (3-vs-3 HighLevelEnv worlds sharded over ranks) on two gloo ranks"""
    n_per_rank, steps, ws = 20, 8, 2
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_hier, args=(ws, port, n_per_rank, steps, str(tmp_path)), nprocs=ws, join=True)
    o = oracle.OracleWorld(oracle.make_config(n_arenas=n_per_rank * ws, env_kind=1, seed=42, auto_reset=True, horizon=48))
    o.reset()
    rng = np.random.default_rng(0)
    cmd_all = rng.integers(0, 3, (steps, n_per_rank * ws, 3)).astype(np.int8)
    tape_all = random_actions(rng, (16, n_per_rank * ws), 6)
    obs, rew, done = _hier_run(o, cmd_all, tape_all)
    ret, ln, oc = o.episode_stats()
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(ws)]
    assert np.array_equal(np.concatenate([x["obs"] for x in r], axis=1), obs)
    assert np.array_equal(np.concatenate([x["rew"] for x in r], axis=1), rew)
    assert np.array_equal(np.concatenate([x["done"] for x in r], axis=1), done)
    want = np.stack([ret, ln.astype(np.float32), oc.astype(np.float32)], axis=1)
    for x in r:
        assert np.array_equal(x["stats"], want)
    assert done.sum() > 0


def test_shard_kwargs_offsets():
    from hhmarl_2d_amd.sharding import shard_kwargs
    kw = dict(n_arenas=8192, seed=1, arena_offset=100)
    assert [shard_kwargs(kw, r, 8)["arena_offset"] for r in range(8)] == [100 + 8192 * r for r in range(8)]
    assert shard_kwargs(kw, 3, 8)["seed"] == 1 and kw["arena_offset"] == 100
