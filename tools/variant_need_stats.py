"""steady-state statistics behind the variant rows: of the alive agents, how many could still raise their weapon flag in a sub-step (flag down AND a way to
raise it: cannon ammunition left, or a type-1 aircraft with a missile left, none in flight, wait over)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hhmarl_2d_amd.env_hier import macro_step
from hhmarl_2d_amd.pilots import VariantNetPilot
from hhmarl_2d_amd.world import World, make_config
N = 2048
w = World(make_config(n_arenas=N, env_kind=1, seed=0, auto_reset=True)); w.reset()
p = VariantNetPilot(w, seed=0)
rng = np.random.default_rng(0)
out = w.alloc_outputs(); pb = w.alloc_pilot_variants()
for step in range(400):
    cmd = torch.from_numpy(rng.integers(0, 3, (N, 3)).astype(np.int8)).cuda()
    macro_step(w, cmd, p, out=out, pilot_buf=pb)
    if step in (50, 200, 399):
        st = w.get_state()
        ai = st["ac_i"][:, :3]   # agents
        alive, typ, crem, burst = ai[..., 0] != 0, ai[..., 1], ai[..., 2], ai[..., 3]
        print(f"step {step}: agents alive {alive.mean():.3f}; of the alive: cannon empty {(crem[alive] == 0).mean():.3f}, burst running {(burst[alive] > 0).mean():.3f}; ac_i columns of one alive agent: {ai[alive][0].tolist()}")
p.close()
