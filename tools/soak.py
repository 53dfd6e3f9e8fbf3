"""One-off soak: the GPU world against the CPU oracle over tens of millions of arena-steps (rare-event coverage for
the staged envelope predicates).  Usage: soak.py [arenas] [ticks] [level] [agent_mode] [esc_dist_rew, default = agent_mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle_lib as O
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
shaping = bool(int(sys.argv[5])) if len(sys.argv) > 5 else bool(mode)
kw = dict(n_arenas=N, level=level, agent_mode=mode, esc_dist_rew=shaping, seed=20260927, auto_reset=True, ext_opp_actions=level >= 4)
if os.environ.get("SOAK_KW"):   # extra configuration fields as JSON, e.g. SOAK_KW='{"friendly_punish": true, "glob_frac": 0.3}' (the general, non-preset kernel instances)
    import json
    kw.update(json.loads(os.environ["SOAK_KW"]))
g = World(make_config(**kw))
o = O.OracleWorld(O.make_config(**kw))
assert np.array_equal(g.reset().cpu().numpy(), o.reset())
rng = np.random.default_rng(99)
hi = np.array([13, 9, 2, 2])
chunk, bad, t0 = 100, 0, time.time()
for c0 in range(0, T, chunk):
    n = min(chunk, T - c0)
    act = (rng.random((n, N, g.n_ctrl, 4)) * hi).astype(np.int8)
    go = [x.cpu().numpy() for x in g.rollout(torch.from_numpy(act).cuda())]
    oo = o.rollout(act)
    for a, b, name in zip(go, oo, ("obs", "reward", "valid", "done")):
        if not np.array_equal(a, b):
            idx = np.argwhere(a != b)
            print(f"MISMATCH {name} chunk {c0}: {len(idx)} entries, first {idx[:3].tolist()}")
            bad += 1
    if bad:
        break
sg, so = g.get_state(), o.get_state()
for k in sg:
    if not np.array_equal(sg[k], so[k]):
        print("MISMATCH final state", k); bad += 1
print(f"{'FAIL' if bad else 'OK'}: {N} arenas x {T} ticks = {N * T / 1e6:.1f} M arena-steps, level {level} mode {mode} {os.environ.get('SOAK_KW', '')} kernel {g.kernel_instance()}, {time.time() - t0:.0f} s")
