"""HighLevelEnv with more than three aircraft on a side (the reference's "any n-vs-m", README.md:43; env_hier.py:226-250 spawns any count):
ten unit slots per arena on the LDS-exchange kernels (hh_k_hier<10, 64, 1> / hh_k_hier_macro<10, ...>), opponents keeping sorted lists of
up to five agents (env_hier.py:97).  Bit for bit against the oracle, which replays the five 4..5-per-side traces recorded from the real
reference (tests/golden/env_hl_fz_{5v5,4v4,5v2,1v4,4v5}*.npz; those run on the GPU in test_gpu_hier.py::test_reference_traces_on_gpu)."""
import numpy as np
import pytest

from helpers import random_actions

pytestmark = pytest.mark.gpu

SIDES = [(5, 5), (4, 2), (1, 5), (5, 3), (4, 4)]


def _same_state(g, o, nU, what):
    for k in g:
        a = g[k] if k == "ar_i" else g[k][:, :nU]
        assert np.array_equal(a, o[k]), f"{what}: {k}"
    assert not g["ac_i"][:, nU:, 0].any(), f"{what}: an unused unit slot is alive"


@pytest.mark.parametrize("sides", SIDES, ids=lambda s: f"{s[0]}v{s[1]}")
def test_wide_phases_against_oracle(oracle, sides):
    """phase by phase (the path the pilot networks run on): every pilot observation row and selector byte of both sides, event masks in the ten-slot
    layout (HH_EV_BIT), running counters, commander outputs, eval counters, full state incl. the five-entry target lists"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    nA, nO = sides
    nU, N = nA + nO, 31   # six arenas per wave: five full groups and one of one
    kw = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=77 + nA, arena_offset=900 + nO, auto_reset=True, horizon=90,
              glob_frac=0.3 if nA == 5 else 0.0, hier_opp_fight_ratio=100 if nO == 2 else 75, friendly_kill=(nA + nO) % 2 == 0)
    g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
    assert g.A == 10 and g.tgt_k == 5
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    _same_state(g.get_state(), o.get_state(), nU, "reset")
    rng = np.random.default_rng(5 * nA + nO)
    dones = kills = picks = 0
    for step in range(14):
        cmd = rng.integers(0, 3, (N, nA)).astype(np.int8)
        po, pm = g.hl_begin(torch.from_numpy(cmd).cuda())
        o.hl_begin(cmd)
        picks += int((o.hl_cmd()[0][:, nA:] >= 4).sum())   # an opponent chose its fourth / fifth nearest agent (randint(2, possible))
        for sub in range(16):
            po_o, pm_o = o.hl_pilot_obs(0)
            assert np.array_equal(pm.cpu().numpy()[:, :nU], pm_o) and np.array_equal(po.cpu().numpy()[:, :nU], po_o), f"{step}/{sub}: agents' pilot rows"
            assert not pm.cpu().numpy()[:, nU:].any() and not po.cpu().numpy()[:, nU:].any()
            act = random_actions(rng, (N,), 10)
            if step % 2 == 0:
                act[..., 2] = 1
            if step % 3 == 0:
                act[..., 3] = 1
            ta = torch.from_numpy(act).cuda()
            ao = np.ascontiguousarray(act[:, :nU])
            po, pm = g.hl_agents_act(ta)
            o.hl_agents_act(ao)
            po_o, pm_o = o.hl_pilot_obs(1)
            assert np.array_equal(pm.cpu().numpy()[:, :nU], pm_o) and np.array_equal(po.cpu().numpy()[:, :nU], po_o), f"{step}/{sub}: opponents' pilot rows"
            po, pm, running = g.hl_tick(ta)
            assert running == o.hl_tick(ao), f"{step}/{sub}: running"
            em = o.event_masks()
            assert np.array_equal(g.event_masks(), em), f"{step}/{sub}: event masks"
            kills += int(np.count_nonzero(em & 0xFFFFF))   # classes 0 / 1 of the ten-slot layout: bits 0..19
            if running == 0:
                break
        for a, b, name in zip([x.cpu().numpy() for x in g.hl_end()], o.hl_end(), ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b), f"step {step}: {name}"
            if name == "done":
                dones += int(a.sum())
        _same_state(g.get_state(), o.get_state(), nU, f"step {step}")
        for a, b, name in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info(), ("last", "total")):
            assert np.array_equal(a, b), f"step {step}: eval_info {name}"
    assert kills > 0 and dones > 0
    if nA == 5:
        assert picks > 0   # the fourth / fifth list entries were exercised


@pytest.mark.parametrize("sides", [(5, 5), (2, 4), (5, 1)], ids=lambda s: f"{s[0]}v{s[1]}")
def test_wide_one_launch_macro_step_against_oracle(oracle, sides):
    """hh_hl_rollout (one launch per commander step, pilot actions from a tape) on ten-slot arenas = the oracle stepping the same tape phase by phase"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    nA, nO = sides
    nU, N = nA + nO, 200
    kw = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=1234, arena_offset=31, auto_reset=True, horizon=120, hier_action_assess=nA != 2)
    g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    rng = np.random.default_rng(99)
    for step in range(10):
        cmd = rng.integers(0, 3, (N, nA)).astype(np.int8)
        tape = random_actions(rng, (16, N), 10)
        tape[..., 2] |= step % 2
        outs = [x.cpu().numpy() for x in g.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())]
        o.hl_begin(cmd)
        for k in range(16):
            o.hl_agents_act(np.ascontiguousarray(tape[k][:, :nU]))
            o.hl_tick(np.ascontiguousarray(tape[k][:, :nU]))
        for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b), (step, name)
        for a, b in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info()):
            assert np.array_equal(a, b), (step, "eval counters")
        assert np.array_equal(g.event_masks(), o.event_masks()), step
        _same_state(g.get_state(), o.get_state(), nU, f"step {step}")


def test_wide_state_round_trip_and_limits(oracle):
    """hh_get_state / hh_set_state carry the five-entry lists (views [N, 10, 5]); a state handed from the oracle continues identically; six per side is refused"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    kw = dict(n_arenas=13, env_kind=1, n_agents=5, n_opps=4, seed=8, arena_offset=0, horizon=200)
    g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
    g.reset(); o.reset()
    rng = np.random.default_rng(3)
    for step in range(3):   # the oracle alone
        o.hl_begin(rng.integers(0, 3, (13, 5)).astype(np.int8))
        for k in range(16):
            a = random_actions(rng, (13,), 9)
            o.hl_agents_act(a); o.hl_tick(a)
        o.hl_end()
    so = o.get_state()
    assert so["tgt_id"].shape == (13, 9, 5) and (so["tgt_id"][:, 5:, 3] > 0).any()   # some opponent lists four agents or more
    sg = g.get_state()
    for k in so:
        if k == "ar_i":
            sg[k][...] = so[k]
        else:
            sg[k][...] = 0
            sg[k][:, :9] = so[k]
    g.set_state(sg)
    _same_state(g.get_state(), so, 9, "after set_state")
    g.observe()   # hh_set_state leaves the commander rows / lists to a refresh: the lists must come back as the oracle has them
    _same_state(g.get_state(), so, 9, "after observe")
    cmd = rng.integers(0, 3, (13, 5)).astype(np.int8)
    tape = random_actions(rng, (16, 13), 10)
    outs = [x.cpu().numpy() for x in g.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())]
    o.hl_begin(cmd)
    for k in range(16):
        o.hl_agents_act(np.ascontiguousarray(tape[k][:, :9])); o.hl_tick(np.ascontiguousarray(tape[k][:, :9]))
    for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
        assert np.array_equal(a, b), name
    with pytest.raises(Exception):
        World(make_config(n_arenas=1, env_kind=1, n_agents=6, n_opps=3))
