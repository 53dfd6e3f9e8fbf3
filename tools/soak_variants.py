"""Soak of round 5's two late additions (run on the GPU box; log -> profiles/r05_soak_late.log):
  (a) the variant-row form of the commander step with the pilot networks in the loop against the two-calls-per-sub-step form: outputs, state, eval
      counters, event masks, tick counts after every commander step, over seeded random configurations (side sizes up to 3, reward sharing, action
      assessment, opponents' fight ratio, horizon, friendly fire) — both banks on one policy kernel form;
  (b) ten-slot arenas (4 - 5 aircraft on a side) against the CPU oracle through the one-launch macro step with a random pilot tape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
os.environ["HH_POLICY_W"] = "0"; os.environ["HH_POLICY_TILE"] = "32"
import numpy as np, torch
from hhmarl_2d_amd.world import World, make_config
from hhmarl_2d_amd.pilots import NetPilot, PolicyBank, VariantNetPilot
from hhmarl_2d_amd.env_hier import macro_step
import oracle_lib as O

rng = np.random.default_rng(20260928)
t0 = time.time()
total = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    nA, nO = (3, 3) if trial % 2 == 0 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    kw = dict(n_arenas=int(rng.choice([257, 1024, 2048])), env_kind=1, n_agents=nA, n_opps=nO, horizon=int(rng.choice([60, 150, 500])),
              glob_frac=float(rng.choice([0.0, 0.3])), hier_action_assess=bool(rng.integers(0, 2)), hier_opp_fight_ratio=int(rng.choice([0, 50, 75, 100])),
              friendly_kill=bool(rng.integers(0, 4)), seed=int(rng.integers(0, 1 << 30)),
              arena_offset=int(rng.integers(0, 1 << 20)), auto_reset=True)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    seed = int(rng.integers(1000))
    pa = NetPilot(a, PolicyBank.random_init(a.device, seed=seed, max_rows=a.N * 6))
    pb = VariantNetPilot(b, PolicyBank.random_init(b.device, seed=seed, max_rows=b.N * 15))
    assert torch.equal(a.reset(), b.reset())
    steps = 120
    for step in range(steps):
        cmd = torch.from_numpy(rng.integers(0, 3, (kw["n_arenas"], nA)).astype(np.int8)).cuda()
        for x, y in zip(macro_step(a, cmd, pa), macro_step(b, cmd, pb)):
            assert torch.equal(x, y), (trial, step, kw)
        if step % 10 == 9:
            sa, sb = a.get_state(), b.get_state()
            assert all(np.array_equal(sa[k], sb[k]) for k in sa), (trial, step, kw)
            assert all(torch.equal(x, y) for x, y in zip(a.eval_info(), b.eval_info())) and np.array_equal(a.event_masks(), b.event_masks())
            assert a.hl_tick_count() == b.hl_tick_count()
    total += steps * kw["n_arenas"]
    print(f"(a) trial {trial}: {nA}v{nO} x {kw['n_arenas']} arenas x {steps} commander steps identical ({a.hl_tick_count()} arena-ticks)  [{time.time() - t0:.0f} s]", flush=True)
    pa.close(); pb.close()
print(f"(a) variant rows = two calls per sub-step on {total} commander steps")

total = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    nA, nO = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    if max(nA, nO) <= 3:
        nA = 4 + trial % 2
    kw = dict(n_arenas=int(rng.choice([37, 150, 301])), env_kind=1, n_agents=nA, n_opps=nO, horizon=int(rng.choice([60, 150])), glob_frac=float(rng.choice([0.0, 0.3])),
              hier_action_assess=bool(rng.integers(0, 2)), hier_opp_fight_ratio=int(rng.choice([0, 50, 75, 100])), friendly_kill=bool(rng.integers(0, 4)),
              seed=int(rng.integers(0, 1 << 30)), arena_offset=int(rng.integers(0, 1 << 20)), auto_reset=True)
    g, o = World(make_config(**kw)), O.OracleWorld(O.make_config(**kw))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    N, nU = kw["n_arenas"], nA + nO
    steps = 25
    for step in range(steps):
        cmd = rng.integers(0, 3, (N, nA)).astype(np.int8)
        tape = np.stack([rng.integers(0, 13, (16, N, 10)), rng.integers(0, 9, (16, N, 10)), rng.integers(0, 2, (16, N, 10)), rng.integers(0, 2, (16, N, 10))], axis=-1).astype(np.int8)
        outs = [x.cpu().numpy() for x in g.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())]
        o.hl_begin(cmd)
        for k in range(16):
            o.hl_agents_act(np.ascontiguousarray(tape[k][:, :nU])); o.hl_tick(np.ascontiguousarray(tape[k][:, :nU]))
        for x, y in zip(outs, o.hl_end()):
            assert np.array_equal(x, y), (trial, step, kw)
        assert np.array_equal(g.event_masks(), o.event_masks()), (trial, step, kw)
    sg, so = g.get_state(), o.get_state()
    assert all(np.array_equal(sg[k] if k == "ar_i" else sg[k][:, :nU], so[k]) for k in sg), (trial, kw)
    total += steps * N
    print(f"(b) trial {trial}: {nA}v{nO} x {N} arenas x {steps} commander steps = the oracle  [{time.time() - t0:.0f} s]", flush=True)
print(f"(b) ten-slot arenas = oracle on {total} commander steps")
