"""Untrusted action words on the GPU: every kernel form that loads action words — the 2-vs-2 register-exchange
kernel in its 8-arenas-per-wave / two-wave / single-wave / two-per-SIMD instances, the generic LDS-exchange kernel of the split step,
the HighLevelEnv phase launches and the one-launch macro step — runs an out-of-range component SANITISED (include/hh_spec.h:
hh_action_sanitize) and raises the arena's sticky fault flag (hh_action_faults): same trajectory bit for bit as the oracle fed the
same words (the oracle sanitises by the same definition, tests/test_action_sanitize.py pins that one against plain clamping), same
flags.  Replaces the reference's raising guards (ac1.py:58-66)."""
import numpy as np
import pytest

from helpers import random_actions
from test_action_sanitize import dirty_actions

pytestmark = pytest.mark.gpu


def _same_state(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: state {k}"


@pytest.mark.parametrize("N,force_w", [(300, "0"), (5000, "0"), (20000, "0"), (700, "2")],
                         ids=["300-8-per-wave", "5000-two-wave", "20000-single-wave", "700-forced-W2"])
def test_rollout_and_step_on_dirty_actions(oracle, monkeypatch, N, force_w):
    import torch
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_FORCE_W", force_w)
    kw = dict(n_arenas=N, level=3, seed=31, arena_offset=2, auto_reset=True, horizon=50)
    g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    rng = np.random.default_rng(N)
    T = 40
    tape = dirty_actions(rng, (T, N), 2, frac=0.02)
    got = [x.cpu().numpy() for x in g.rollout(torch.from_numpy(tape).cuda())]
    for x, y, name in zip(got, o.rollout(tape), ("obs", "reward", "valid", "done")):
        assert np.array_equal(x, y), f"rollout: {name}"
    _same_state(g.get_state(), o.get_state(), "after the rollout")
    f = g.action_faults().cpu().numpy()
    assert np.array_equal(f, o.action_faults()) and f.any() and not f.all()
    assert not f[np.arange(N) % 3 != 0].any(), "arenas whose words were in range stay clean"
    # sticky across steps and resets; cleared on request only
    for t in range(6):
        act = random_actions(rng, (N,), 2)
        got = [x.cpu().numpy() for x in g.step(torch.from_numpy(act).cuda())]
        for x, y in zip(got, o.step(act)):
            assert np.array_equal(x, y)
    assert np.array_equal(g.action_faults(clear=True).cpu().numpy(), f)
    assert not g.action_faults().cpu().numpy().any()
    o.action_faults(clear=True)
    act = dirty_actions(rng, (N,), 2, frac=0.5, every=2)
    got = [x.cpu().numpy() for x in g.step(torch.from_numpy(act).cuda())]
    for x, y in zip(got, o.step(act)):
        assert np.array_equal(x, y)
    f2 = g.action_faults().cpu().numpy()
    assert np.array_equal(f2, o.action_faults()) and f2.any()


def test_split_step_on_dirty_actions(oracle):
    import torch
    from hhmarl_2d_amd.world import World, make_config
    N = 400
    kw = dict(n_arenas=N, level=4, seed=17, auto_reset=True, ext_opp_actions=True, horizon=40)
    g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    rng = np.random.default_rng(3)
    for t in range(60):
        act = dirty_actions(rng, (N,), 4, frac=0.01)
        a_ag, a_op = np.ascontiguousarray(act[:, :2]), np.ascontiguousarray(act[:, 2:])
        oo = g.step_begin(torch.from_numpy(a_ag).cuda(), 0).cpu().numpy()
        assert np.array_equal(oo, o.step_begin(a_ag, 0)), f"t={t}: opponents' observations"
        outs = [x.cpu().numpy() for x in g.step_finish(torch.from_numpy(a_op).cuda())]
        for x, y, name in zip(outs, o.step_finish(a_op), ("obs", "reward", "valid", "done")):
            assert np.array_equal(x, y), f"t={t}: {name}"
    _same_state(g.get_state(), o.get_state(), "final")
    f = g.action_faults().cpu().numpy()
    assert np.array_equal(f, o.action_faults()) and f.any() and not f.all()


@pytest.mark.parametrize("N,no_oct", [(170, "0"), (170, "1"), (9000, "0")], ids=["170-register-exchange", "170-lds-exchange", "9000-W2"])
def test_commander_step_on_dirty_pilot_actions(oracle, monkeypatch, N, no_oct):
    """phase launches against the oracle (N = 170), and the one-launch macro step against the phase launches, on a tape with out-of-range words"""
    import torch
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_NO_OCT", no_oct)
    base = dict(n_arenas=N, env_kind=1, seed=8, arena_offset=11, auto_reset=True, horizon=60)
    a, b = World(make_config(**base)), World(make_config(**base))
    o = oracle.OracleWorld(oracle.make_config(**base)) if N <= 200 else None
    assert torch.equal(a.reset(), b.reset())
    if o is not None:
        o.reset()
    rng = np.random.default_rng(N)
    for step in range(6):
        cmd_h = rng.integers(0, 3, (N, 3)).astype(np.int8)
        tape_h = dirty_actions(rng, (16, N), 6, frac=0.01)
        cmd, tape = torch.from_numpy(cmd_h).cuda(), torch.from_numpy(tape_h).cuda()
        calls = [0]

        def pilot(po, pm):
            act = tape[(calls[0] // 2) % 16]
            calls[0] += 1
            return act
        outs_a = macro_step(a, cmd, pilot)
        outs_b = b.hl_rollout(cmd, tape)
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        _same_state(a.get_state(), b.get_state(), f"step {step}")
        if o is not None:
            o.hl_begin(cmd_h)
            for k in range(16):
                o.hl_agents_act(tape_h[k])
                o.hl_tick(tape_h[k])
            for x, y, name in zip([t.cpu().numpy() for t in outs_b], o.hl_end(), ("obs", "reward", "valid", "done")):
                assert np.array_equal(x, y), f"step {step}: {name} vs oracle"
            _same_state(b.get_state(), o.get_state(), f"step {step} vs oracle")
    fa, fb = a.action_faults().cpu().numpy(), b.action_faults().cpu().numpy()
    assert np.array_equal(fa, fb) and fa.any() and not fa.all()
    if o is not None:
        assert np.array_equal(fa, o.action_faults())
