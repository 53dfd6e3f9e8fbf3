"""One-off soak of the HighLevelEnv macro step, GPU against the CPU oracle.  Usage: soak_hier.py [arenas] [commander steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle_lib as O
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8195
S = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kw = dict(n_arenas=N, env_kind=1, seed=77, arena_offset=12345, auto_reset=True)
g = World(make_config(**kw)); o = O.OracleWorld(O.make_config(**kw))
assert np.array_equal(g.reset().cpu().numpy(), o.reset())
rng = np.random.default_rng(8)
hi = np.array([13, 9, 2, 2])
ticks, t0 = 0, time.time()
for step in range(S):
    cmd = rng.integers(0, 3, (N, 3)).astype(np.int8)
    po, pm = g.hl_begin(torch.from_numpy(cmd).cuda()); o.hl_begin(cmd)
    for sub in range(16):
        po_o, pm_o = o.hl_pilot_obs(0)
        assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), (step, sub, "agents' pilot obs")
        act = (rng.random((N, 6, 4)) * hi).astype(np.int8)
        if step % 3 == 0:
            act[..., 2] = 1
        ta = torch.from_numpy(act).cuda()
        po, pm = g.hl_agents_act(ta); o.hl_agents_act(act)
        po_o, pm_o = o.hl_pilot_obs(1)
        assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), (step, sub, "opponents' pilot obs")
        po, pm, running = g.hl_tick(ta)
        assert running == o.hl_tick(act), (step, sub, "running")
        assert np.array_equal(g.event_masks(), o.event_masks()), (step, sub, "event masks")
        ticks += running
        if running == 0:
            break
    for a, b, name in zip([x.cpu().numpy() for x in g.hl_end()], o.hl_end(), ("obs", "reward", "valid", "done")):
        assert np.array_equal(a, b), (step, name)
sg, so = g.get_state(), o.get_state()
for k in sg:
    assert np.array_equal(sg[k], so[k]), k
print(f"OK: {N} arenas x {S} commander steps (~{ticks / 1e6:.1f} M arena-ticks), {time.time() - t0:.0f} s")
