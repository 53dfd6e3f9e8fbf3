"""how many rows the variant-row form lists per sub-step (and how they split into agents' rows / opponents' base rows / variants), 8192 arenas, random-init networks"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hhmarl_2d_amd.world import World, make_config
from hhmarl_2d_amd.pilots import VariantNetPilot
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = World(make_config(n_arenas=N, env_kind=1, seed=0, auto_reset=True)); w.reset()
p = VariantNetPilot(w, seed=0)
rng = np.random.default_rng(0)
tot = np.zeros(4); subs = 0
for step in range(40):
    cmd = torch.from_numpy(rng.integers(0, 3, (N, 3)).astype(np.int8)).cuda()
    po, pm = w.hl_begin_variants(cmd)
    for sub in range(16):
        m = (pm != 0).cpu().numpy()
        if step >= 20:
            o = m[:, 3:].reshape(N, 3, 4)
            tot += [m[:, :3].sum(), o[:, :, 0].sum(), o[:, :, 1:].sum(), (o.sum(-1) == 4).sum()]
            subs += 1
        po, pm, r = w.hl_act_tick(p(po, pm), count_running=False)
    w.hl_end()
print("per sub-step of %d arenas: agents' rows %.0f, opponents' base rows %.0f, variant rows %.0f (opponents with all four: %.0f); total %.0f = %.2f per arena" % (
    N, *(tot / subs), tot[:3].sum() / subs, tot[:3].sum() / subs / N))
