"""One-off soak of the one-launch HighLevelEnv macro step (hh_hl_rollout, every instance the arena count selects) against the CPU
oracle stepping the same tape phase by phase.  Usage: soak_macro.py [arenas] [commander steps] [horizon]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle_lib as O
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H = int(sys.argv[3]) if len(sys.argv) > 3 else 500
kw = dict(n_arenas=N, env_kind=1, seed=91, arena_offset=777, auto_reset=True, horizon=H)
g = World(make_config(**kw)); o = O.OracleWorld(O.make_config(**kw))
assert np.array_equal(g.reset().cpu().numpy(), o.reset())
rng = np.random.default_rng(5)
hi = np.array([13, 9, 2, 2])
t0, dones = time.time(), 0
for step in range(S):
    cmd = rng.integers(0, 3, (N, 3)).astype(np.int8)
    tape = (rng.random((16, N, 6, 4)) * hi).astype(np.int8)
    if step % 3 == 0:
        tape[..., 2] = 1
    outs = [x.cpu().numpy() for x in g.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())]
    o.hl_begin(cmd)
    for k in range(16):
        o.hl_agents_act(tape[k])
        o.hl_tick(tape[k])
    for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
        assert np.array_equal(a, b), (step, name)
    for a, b in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info()):
        assert np.array_equal(a, b), (step, "eval counters")
    dones += int(outs[3].sum())
sg, so = g.get_state(), o.get_state()
for k in sg:
    assert np.array_equal(sg[k], so[k]), k
print(f"OK: {N} arenas x {S} commander steps in one launch each ({g.hl_tick_count() / 1e6:.1f} M arena-ticks, {dones} episodes ended), {time.time() - t0:.0f} s")
