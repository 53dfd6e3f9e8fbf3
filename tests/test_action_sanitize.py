"""Untrusted action words (the reference's raising guards ac1.py:58-66): a component outside
MultiDiscrete([13, 9, 2, 2]) is SANITISED where the word is consumed (include/hh_spec.h: hh_action_sanitize — heading component
clamped to [0, 12], speed component to [0, 8], fire components read as booleans) and the arena's sticky fault flag is raised.
CPU half: the oracle's definition (the GPU half, tests/test_gpu_parity.py, holds the kernels to it)."""
import numpy as np
import pytest

from helpers import random_actions


def sanitised(act):
    """what hh_action_sanitize makes of int8 [..., 4] action words"""
    out = act.copy()
    out[..., 0] = np.clip(act[..., 0], 0, 12)
    out[..., 1] = np.clip(act[..., 1], 0, 8)
    out[..., 2] = act[..., 2] != 0
    out[..., 3] = act[..., 3] != 0
    return out


def dirty_actions(rng, shape_prefix, n_ctrl, frac=0.05, every=3):
    """uniform in-range actions; in every `every`-th arena a fraction of the words has one component replaced by an arbitrary int8"""
    a = random_actions(rng, shape_prefix, n_ctrl)
    bad = rng.random(a.shape[:-1]) < frac
    bad &= (np.arange(a.shape[-3]) % every == 0)[:, None]
    junk = rng.integers(-128, 128, size=a.shape).astype(np.int8)
    comp = rng.integers(0, 4, size=a.shape[:-1])
    for k in range(4):
        sel = bad & (comp == k)
        a[..., k][sel] = junk[..., k][sel]
    return a


@pytest.mark.parametrize("kw", [dict(level=3), dict(level=1, agent_mode=1), dict(level=4, ext_opp_actions=True)],
                         ids=["L3-fight", "L1-escape", "L4-ext-opp"])
def test_oracle_runs_out_of_range_actions_sanitised_and_flags_the_arena(oracle, kw):
    N, T = 48, 60
    cfg = dict(n_arenas=N, seed=21, arena_offset=7, auto_reset=True, horizon=40, **kw)
    a, b = oracle.OracleWorld(oracle.make_config(**cfg)), oracle.OracleWorld(oracle.make_config(**cfg))
    assert np.array_equal(a.reset(), b.reset())
    rng = np.random.default_rng(4)
    expect = np.zeros(N, dtype=np.uint8)
    for t in range(T):
        act = dirty_actions(rng, (N,), a.n_ctrl)
        st = a.get_state()
        alive = st["ac_i"][:, :a.n_ctrl, 0] != 0                        # consumed rows: live units (every arena runs: auto-reset)
        clean = sanitised(act)
        expect |= ((clean != act).any(-1) & alive).any(-1).astype(np.uint8)
        got, want = a.step(act), b.step(clean)
        for x, y, name in zip(got, want, ("obs", "reward", "valid", "done")):
            assert np.array_equal(x, y), f"t={t}: {name}"
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert expect.any() and not expect.all()
    assert np.array_equal(a.action_faults(), expect)
    assert not b.action_faults().any(), "in-range actions never raise the flag"
    assert np.array_equal(a.action_faults(clear=True), expect) and not a.action_faults().any(), "sticky until cleared"


def test_oracle_commander_step_sanitises_pilot_actions(oracle):
    N = 24
    cfg = dict(n_arenas=N, env_kind=1, seed=5, auto_reset=True, horizon=30)
    a, b = oracle.OracleWorld(oracle.make_config(**cfg)), oracle.OracleWorld(oracle.make_config(**cfg))
    a.reset(); b.reset()
    rng = np.random.default_rng(8)
    for step in range(6):
        cmd = rng.integers(0, 3, (N, 3)).astype(np.int8)
        a.hl_begin(cmd); b.hl_begin(cmd)
        for sub in range(16):
            act = dirty_actions(rng, (N,), 6, frac=0.1)
            clean = sanitised(act)
            a.hl_agents_act(act); b.hl_agents_act(clean)
            ra, rb = a.hl_tick(act), b.hl_tick(clean)
            assert ra == rb
            if ra == 0:
                break
        for x, y in zip(a.hl_end(), b.hl_end()):
            assert np.array_equal(x, y)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert a.action_faults().any() and not b.action_faults().any()
