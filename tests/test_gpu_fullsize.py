"""-m gpu: one independent look at BASELINE size.  The bench's own world — 4096 arenas x 2-vs-2 fight level 3, seed 1234, auto-reset, the
kernel instance bench.py times — runs 300 ticks; sixteen of its arenas (first, last, both sides of workgroup boundaries) are compared with
trajectories recorded from the REAL reference (libm-based geodesic, nothing of the shared headers: tests/golden/fullsize_l3.npz), so a
defect in include/hh_math.h / hh_geodesic.h / hh_rng.h cannot hide behind the HIP-vs-oracle comparisons at this size."""
import json

import numpy as np
import pytest
import torch

from test_fullsize_oracle import GOLD, compare_arena, tape_of


@pytest.mark.gpu
@pytest.mark.parametrize("n_world", [4096, 16384])
def test_sixteen_arenas_of_the_full_world_match_the_reference(n_world):
    from hhmarl_2d_amd.world import World, make_config
    g = np.load(GOLD)
    meta = json.loads(str(g["meta"]))
    T, chunk, arenas = meta["ticks"], meta["chunk"], [int(a) for a in g["arenas"]]
    w = World(make_config(n_arenas=n_world, level=meta["level"], seed=meta["seed"], auto_reset=True), device=0)
    if n_world == 4096:
        assert "hh_k_world_quad" in w.kernel_name()            # the instance the headline is measured on
    w.reset()
    rng = np.random.default_rng(99)
    tape = rng.integers(0, [13, 9, 2, 2], (T, n_world, 2, 4)).astype(np.int8)     # every other arena: anything
    for a in arenas:
        tape[:, a] = tape_of(meta["seed"], a, T)
    tape = torch.from_numpy(tape).cuda()
    outs, snaps = [], {a: [] for a in arenas}
    for c in range(T // chunk):
        o, r, v, d = w.rollout(tape[c * chunk:(c + 1) * chunk].contiguous())
        outs.append([x.cpu().numpy() for x in (o, r, v, d)])
        st = w.get_state()
        for a in arenas:
            snaps[a].append((st["ac_f"][a], st["ac_i"][a], st["ar_i"][a]))
    obs, rew, val, done = (np.concatenate([x[i] for x in outs]) for i in range(4))
    for k, a in enumerate(arenas):
        compare_arena(g, k, obs[:, a], rew[:, a], val[:, a], done[:, a], snaps[a], f"arena {a} of {n_world}")
