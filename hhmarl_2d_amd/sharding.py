"""Arena sharding across GPUs: one process per GPU, each rank owns a complete
world of n_arenas arenas with global ids [rank*n_arenas, (rank+1)*n_arenas).  Stepping needs NO
communication (arenas are independent and the keyed RNG is indexed by the global arena id, so a
sharded run is bit-identical to one big world).  The only exchange is logging: an all-gather of
per-arena episode statistics ([n_arenas, 3] float32 = return, length, outcome) over RCCL/xGMI
(`torch.distributed` backend "nccl"), or gloo in the CPU tests."""
import torch


def shard_kwargs(cfg_kwargs, rank, world_size):
    """config kwargs of this rank's world: same seed, disjoint global arena ids."""
    kw = dict(cfg_kwargs)
    base = int(kw.get("arena_offset", 0))
    kw["arena_offset"] = base + rank * int(kw["n_arenas"])
    assert 0 <= rank < world_size
    return kw


def pack_stats(ret, length, outcome):
    """[N,3] float32 block moved by the all-gather (96 KB per rank at 8192 arenas)."""
    return torch.stack([ret.float(), length.float(), outcome.float()], dim=1).contiguous()


def gather_stats(block, world_size, group=None):
    """all-gather the per-rank [N,3] blocks -> [world_size*N, 3] in global arena order."""
    import torch.distributed as dist
    if world_size == 1 and not (dist.is_available() and dist.is_initialized()):
        return block
    seen = dist.get_world_size(group)
    if seen != world_size:   # a launcher that started fewer ranks than the shards were cut for: the "global" block would silently lack arenas
        raise RuntimeError(f"gather_stats: the process group has {seen} ranks but the world was sharded over {world_size}")
    out = torch.empty((world_size * block.shape[0], block.shape[1]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block, group=group)
    if out.shape[0] != world_size * block.shape[0]:
        raise RuntimeError(f"gather_stats: {out.shape[0]} rows gathered, {world_size} blocks of {block.shape[0]} expected")
    return out


def gather_evidence(arena_offset, n_arenas, world_size, device, group=None):
    """What a multi-rank bench line quotes so that "did the collective see N ranks" can be read from the line itself: every rank
    contributes (rank, first global arena, arenas) through the SAME backend the statistics travel on (RCCL on GPUs, gloo in the CPU
    tests); the blocks must arrive in rank order and tile the global arena range without gaps."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks_seen": 1, "first_global_arena_of_each_block": [int(arena_offset)], "arenas_of_each_block": [int(n_arenas)], "backend": None}
    seen = dist.get_world_size(group)
    mine = torch.tensor([dist.get_rank(group), int(arena_offset), int(n_arenas)], dtype=torch.int64, device=device)
    out = torch.empty((seen * 3,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    rows = out.view(seen, 3).cpu().tolist()
    if seen != world_size or [r[0] for r in rows] != list(range(seen)):
        raise RuntimeError(f"gather_evidence: {seen} ranks answered {rows}, {world_size} launched")
    for a, b in zip(rows, rows[1:]):
        if b[1] != a[1] + a[2]:
            raise RuntimeError(f"gather_evidence: the ranks' arena ranges do not tile the global range: {rows}")
    return {"ranks_seen": seen, "first_global_arena_of_each_block": [r[1] for r in rows], "arenas_of_each_block": [r[2] for r in rows],
            "backend": dist.get_backend(group)}


def summarize(all_stats):
    """win/lose/draw counts and mean return over finished episodes (outcome 2 = none yet)."""
    oc = all_stats[:, 2]
    fin = oc != 2
    n = int(fin.sum())
    return {"episodes": n, "agents_win": int((oc == 1).sum()), "opps_win": int((oc == -1).sum()),
            "draw": int(((oc == 0) & fin).sum()),
            "mean_return": float(all_stats[fin, 0].mean()) if n else 0.0,
            "mean_length": float(all_stats[fin, 1].mean()) if n else 0.0}


class ShardedWorld:
    """This rank's shard of a multi-GPU world + the logging collective."""

    def __init__(self, cfg_kwargs, rank=0, world_size=1, device=0, world_factory=None):
        self.rank, self.world_size = rank, world_size
        kw = shard_kwargs(cfg_kwargs, rank, world_size)
        if world_factory is None:
            from .world import World, make_config
            self.world = World(make_config(**kw), device=device)
        else:
            self.world = world_factory(kw)
        self.arena_offset, self.n_arenas = int(kw["arena_offset"]), int(kw["n_arenas"])
        self.last_stats = None
        self._blocks, self._events, self._k = [None, None], [None, None], 0

    def log_episode_stats(self, side_stream=None):
        """Logging only.  The [N,3] block is snapshotted by ONE small launch on the stepping stream
        (hh_episode_stats_packed); with `side_stream` the all-gather then runs there, so the stepping stream goes straight
        on to the next rollout launch — call it every K launches, not every launch.  Blocks are double-buffered: a
        snapshot waits only for the gather that used the same buffer two calls ago."""
        w = self.world
        if not hasattr(w, "episode_stats_packed"):          # CPU test doubles
            self.last_stats = gather_stats(pack_stats(*w.episode_stats()), self.world_size)
            return self.last_stats
        k = self._k = self._k ^ 1
        cur = torch.cuda.current_stream(w.device)
        if self._events[k] is not None:
            cur.wait_event(self._events[k])
        self._blocks[k] = w.episode_stats_packed(self._blocks[k])
        if side_stream is None:
            self.last_stats = gather_stats(self._blocks[k], self.world_size)
            return self.last_stats
        side_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            self.last_stats = gather_stats(self._blocks[k], self.world_size)
            self._events[k] = torch.cuda.Event()
            self._events[k].record(side_stream)
        # the gathered tensor was allocated and is being written on the side stream: whoever reads it from the stepping stream
        # must wait for the gather (stats_ready) and the allocator must not recycle the block under a reader on that stream
        self.last_stats.record_stream(cur)
        self.stats_ready = self._events[k]
        return self.last_stats

    def evidence(self):
        """ranks_seen / block order / gathered rows of the logging collective, for the bench line (one tiny all-gather on the current stream)"""
        w = self.world
        ev = gather_evidence(self.arena_offset, self.n_arenas, self.world_size, getattr(w, "device", torch.device("cpu")))
        last = self.wait_stats() if hasattr(w, "episode_stats_packed") else self.last_stats
        ev["gathered_rows"] = None if last is None else int(last.shape[0])
        if last is not None and ev["gathered_rows"] != sum(ev["arenas_of_each_block"]):
            raise RuntimeError(f"logging all-gather returned {ev['gathered_rows']} rows for blocks {ev['arenas_of_each_block']}")
        return ev

    def wait_stats(self):
        """last_stats, safe to read from the current stream (orders it after the side-stream all-gather that produces it)"""
        ev = getattr(self, "stats_ready", None)
        if ev is not None:
            torch.cuda.current_stream(self.world.device).wait_event(ev)
        return self.last_stats
