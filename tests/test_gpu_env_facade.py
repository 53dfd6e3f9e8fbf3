"""The drop-in boundary: hhmarl_2d_amd.env_hetero.LowLevelEnv driven exactly like RLlib drives the
reference's LowLevelEnv (dict in / dict out), checked against the golden traces recorded from the
reference: same keys, same shapes/dtypes, same values."""
import os

import numpy as np
import pytest

from helpers import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _env_from_meta(meta, **extra):
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    a = meta["args"]
    args = make_args(0, **{k: a[k] for k in ("level", "agent_mode", "horizon", "map_size", "glob_frac", "rew_scale",
                                              "esc_dist_rew", "friendly_kill", "friendly_punish")})
    cfg = {"args": args, "seed": meta["seed"]}
    cfg.update(extra)
    env = LowLevelEnv(cfg)
    return env, args


@pytest.mark.parametrize("path", golden_files()[:4], ids=lambda p: p.split("env_")[-1][:-4])
def test_dict_protocol_matches_reference_trace(path):
    g, meta = load_golden(path)
    # the facade numbers arenas from 0; the trace was recorded for global arena id meta["arena"]
    from hhmarl_2d_amd import env_hetero
    orig = env_hetero.config_from_args
    env_hetero.config_from_args = lambda *a, **k: orig(*a, **{**k, "arena_offset": meta["arena"]})
    try:
        env, args = _env_from_meta(meta)
    finally:
        env_hetero.config_from_args = orig
    dims = env.obs_dim_map
    assert env._agent_ids == {1, 2} and env._skip_env_checking
    assert env.observation_space[1].shape == (dims[1],) and env.action_space[2].nvec.tolist() == [13, 9, 2]
    for r in range(len(g["kind"])):
        if g["kind"][r] == 0:
            obs, info = env.reset()
            assert info == {}
        else:
            act = {1: g["actions"][r][0, :4].tolist(), 2: g["actions"][r][1, :3].tolist()}
            obs, rew, term, trunc, info = env.step(act)
            assert term is trunc and set(term) == {"__all__"} and info == {}
            assert term["__all__"] == bool(g["done"][r])
            assert set(rew) == {i + 1 for i in range(2) if g["valid"][r][i]}
            for i in rew:
                assert abs(rew[i] - g["reward"][r][i - 1]) <= 1e-6 * max(1.0, abs(g["reward"][r][i - 1]))
        assert set(obs) == {1, 2}
        for i in (1, 2):
            assert obs[i].dtype == np.float32 and obs[i].shape == (dims[i],)
            assert np.abs(obs[i] - g["obs"][r][i - 1, : dims[i]]).max() <= 1e-6
    env.close()


def test_vector_facade_and_empty_action():
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    env = LowLevelEnv({"args": make_args(0, level=3), "num_envs": 32, "seed": 9})
    obs, _ = env.reset()
    assert obs[1].shape == (32, 26) and obs[2].shape == (32, 24)
    o2, rew, term, trunc, _ = env.step({})
    assert rew == {} and np.array_equal(o2[1], obs[1]) and not term["__all__"].any()
    act = {1: np.tile([6, 4, 1, 1], (32, 1)), 2: np.tile([6, 4, 1], (32, 1))}
    o3, rew, term, _, _ = env.step(act)
    assert o3[1].shape == (32, 26) and rew[1].shape == (32,)
    env.close()


def test_level4_requires_opponent_policy_and_replays_reference_trace():
    import torch
    from hhmarl_2d_amd import env_hetero
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    with pytest.raises(ValueError):
        LowLevelEnv({"args": make_args(0, level=4)})
    path = [p for p in golden_files() if "l4_fight_frozen" in p][0]
    g, meta = load_golden(path)
    cur = {"r": 0}

    def frozen(opp_obs, env):  # the "frozen policy": replays the recorded opponent actions, checks its input
        r = cur["r"]
        assert np.abs(opp_obs.cpu().numpy()[0] - g["opp_obs"][r]).max() <= 1e-6
        return torch.from_numpy(np.ascontiguousarray(g["actions"][r][None, 2:])).to(opp_obs.device)

    orig = env_hetero.config_from_args
    env_hetero.config_from_args = lambda *a, **k: orig(*a, **{**k, "arena_offset": meta["arena"]})
    try:
        env = LowLevelEnv({"args": make_args(0, level=4), "seed": meta["seed"], "opponent_policy": frozen})
    finally:
        env_hetero.config_from_args = orig
    for r in range(120):
        cur["r"] = r
        if g["kind"][r] == 0:
            obs, _ = env.reset()
        else:
            obs, rew, term, _, _ = env.step({1: g["actions"][r][0, :4].tolist(), 2: g["actions"][r][1, :3].tolist()})
            assert term["__all__"] == bool(g["done"][r])
        for i in (1, 2):
            assert np.abs(obs[i] - g["obs"][r][i - 1, : env.obs_dim_map[i]]).max() <= 1e-6
    env.close()


def test_level5_facade_draws_the_opponent_policy_per_episode():
    """envs/env_hetero.py:55-59: at level 5 reset() draws k = randint(3,5) and the opponents observe in escape mode iff k == 5.
    The facade is NOT fed the recorded opp_mode: it must arrive at it through the world's keyed draw, episode after episode."""
    import torch
    from hhmarl_2d_amd import env_hetero
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    path = [p for p in golden_files() if "l5_fight_frozen" in p][0]
    g, meta = load_golden(path)
    cur = {"r": 0}
    seen = set()

    def frozen(opp_obs, env):
        r = cur["r"]
        assert env.opp_mode == ("escape" if g["opp_mode"][r] == 1 else "fight") and env.opp_k in (3, 4, 5)
        assert (env.opp_k == 5) == (env.opp_mode == "escape")
        seen.add(env.opp_mode)
        assert np.abs(opp_obs.cpu().numpy()[0] - g["opp_obs"][r]).max() <= 1e-6, f"row {r}: opponents' observation"
        return torch.from_numpy(np.ascontiguousarray(g["actions"][r][None, 2:])).to(opp_obs.device)

    orig = env_hetero.config_from_args
    env_hetero.config_from_args = lambda *a, **k: orig(*a, **{**k, "arena_offset": meta["arena"]})
    try:
        env = LowLevelEnv({"args": make_args(0, level=5, horizon=meta["args"]["horizon"]), "seed": meta["seed"], "opponent_policy": frozen})
    finally:
        env_hetero.config_from_args = orig
    for r in range(len(g["kind"])):
        cur["r"] = r
        if g["kind"][r] == 0:
            obs, _ = env.reset()
        else:
            obs, rew, term, _, _ = env.step({1: g["actions"][r][0, :4].tolist(), 2: g["actions"][r][1, :3].tolist()})
            assert term["__all__"] == bool(g["done"][r])
        for i in (1, 2):
            assert np.abs(obs[i] - g["obs"][r][i - 1, : env.obs_dim_map[i]]).max() <= 1e-6
    assert seen == {"fight", "escape"}, "the trace must cover both draws"
    env.close()


@pytest.mark.parametrize("which", ["hl_random_pilots", "hl_eval_info", "hl_2v3_eval", "hl_3v1_eval"])
def test_highlevel_dict_protocol_matches_reference_trace(which):
    """HighLevelEnv facade driven like RLlib / evaluation.py drive the reference, with the trace's taped pilot
    actions; with eval_info the info dict (env_base.py:91-107) must equal the reference's"""
    import json
    import torch
    from hhmarl_2d_amd import env_hetero
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    g, meta = load_golden([p for p in golden_files("high") if which in p][0])
    infos = json.loads(str(g["infos"])) if "infos" in g.files else None
    a = meta["args"]
    args = make_args(1, **{k: a[k] for k in ("horizon", "map_size", "glob_frac", "rew_scale", "friendly_kill", "num_agents", "num_opps",
                                              "hier_action_assess", "hier_opp_fight_ratio", "level", "eval_info")})
    nA = a["num_agents"]
    tape = {"ptr": 0}

    def pilot(po, pm):  # replays the recorded pilot actions; called for agents then opponents in each sub-step
        k = tape["ptr"] // 2
        tape["ptr"] += 1
        a6 = np.zeros((1, 6, 4), dtype=np.int8)                 # the world has six unit slots; n-vs-m traces fill the first n + m
        sa = g["sub_act"][min(k, len(g["sub_act"]) - 1)]
        a6[0, : sa.shape[0]] = sa
        return torch.from_numpy(a6).to(po.device)

    orig = env_hetero.config_from_args
    import hhmarl_2d_amd.env_hier as eh
    eh.config_from_args = lambda *x, **k: orig(*x, **{**k, "arena_offset": meta["arena"]})
    try:
        env = HighLevelEnv({"args": args, "seed": meta["seed"], "pilot": pilot})
    finally:
        eh.config_from_args = orig
    assert env._agent_ids == set(range(1, nA + 1)) and env.observation_space.shape == (34,) and env.action_space.n == 3
    for r in range(len(g["kind"])):
        if g["kind"][r] == 0:
            obs, info = env.reset()
        else:
            obs, rew, term, trunc, info = env.step({i + 1: int(g["cmd"][r][i]) for i in range(nA)})
            assert term is trunc and term["__all__"] == bool(g["done"][r]) and set(rew) == set(range(1, nA + 1))
            if infos is not None:
                assert info == infos[r], f"row {r}: eval info {info} != {infos[r]}"
            for i in rew:
                assert abs(rew[i] - g["reward"][r][i - 1]) <= 1e-6
        for i in range(1, nA + 1):
            assert obs[i].dtype == np.float32 and obs[i].shape == (34,)
            assert np.abs(obs[i] - g["obs"][r][i - 1]).max() <= 1e-6
    assert tape["ptr"] == 2 * len(g["sub_act"])
    env.close()


def test_highlevel_vector_facade_eval_info_for_many_arenas():
    """eval_info is counted on the device for every arena (no host copy of the world, no num_envs == 1 restriction)"""
    import torch
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    from hhmarl_2d_amd.pilots import RandomPilot
    n = 64
    env = HighLevelEnv({"args": make_args(1, eval_info=True, horizon=60), "num_envs": n, "seed": 4, "pilot": RandomPilot(torch.device("cuda", 0), 3)})
    obs, _ = env.reset()
    assert obs[1].shape == (n, 34)
    o2, rew, term, _, info = env.step({})                       # empty action dict: observation only, nothing counted
    assert rew == {} and info == {} and not term["__all__"].any()
    sums = None
    for t in range(6):
        obs, rew, term, _, info = env.step({1: np.full(n, t % 3), 2: np.ones(n, dtype=int), 3: np.zeros(n, dtype=int)})
        assert set(info) == {"agents_win", "opps_win", "draw", "agent_fight", "agent_escape", "opp_fight", "opp_escape", "agent_steps",
                             "opp_steps", "opp1", "opp2", "opp3"} and info["agent_steps"].shape == (n,)
        assert (info["agent_fight"] + info["agent_escape"] == info["agent_steps"]).all()
        assert (info["opp1"] + info["opp2"] + info["opp3"] == info["agent_fight"]).all() and (info["agent_steps"] <= 3).all()
        assert ((info["agents_win"] + info["opps_win"] + info["draw"]) <= term["__all__"]).all()   # set on the step that ends the episode
        cur = np.stack([info[k] for k in sorted(info)], axis=1)
        sums = cur if sums is None else sums + cur
    last, tot = env.world.eval_info()
    from hhmarl_2d_amd._lib import EVAL_KEYS
    order = [EVAL_KEYS.index(k) for k in sorted(EVAL_KEYS)]
    assert np.array_equal(tot.cpu().numpy()[:, order], sums)
    env.close()


def test_plot_writes_a_png(tmp_path):
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    env = LowLevelEnv({"args": make_args(0, level=3), "seed": 2, "record_trace": True})
    env.reset()
    for _ in range(30):
        env.step({1: [6, 5, 1, 1], 2: [7, 5, 1]})
    out = tmp_path / "arena.png"
    env.plot(out)
    assert out.exists() and out.stat().st_size > 5000
    # the trajectory comes from the device-side ring buffer and equals the world's state history
    rows, ep = env.world.trace_read()[0]
    assert len(rows) == 31 and (ep == 1).all()                   # reset row + 30 ticks of episode 1
    st = env.world.get_state()
    assert np.allclose(rows[-1, :, :4], st["ac_f"][0, :, :4].astype(np.float32)) and np.array_equal(rows[-1, :, 4], st["ac_i"][0, :, 0])
    assert np.array_equal(rows[-1, :, 7], st["rk_i"][0, :, 0])
    env.close()


def test_trace_ring_equals_the_reference_simulators_own_unit_trace():
    """SURVEY.md 8 f-4 against the reference itself: tests/golden/unit_trace_l3.npz is CmanoSimulator.trace_record_units
    (cmano_simulator.py:125-130,147-150,159-162: (utc_time, position, heading, speed) per aircraft at reset and after every tick
    while it exists) dumped at the end of each episode of the `l3_fight_random` scenario.  Replaying that scenario's recorded
    actions through the facade with record_trace must leave the same trajectory in the device-side ring buffer: same number of
    points per aircraft, same values (float32 ring), and the ring's alive flag drops exactly where the reference stops recording
    (a unit that leaves the map is still recorded at the tick it left: do_tick stores, then the env removes it)."""
    from hhmarl_2d_amd import env_hetero
    g, meta = load_golden([p for p in golden_files() if p.endswith("env_l3_fight_random.npz")][0])
    ref = np.load(os.path.join(os.path.dirname(golden_files()[0]), "unit_trace_l3.npz"))
    orig = env_hetero.config_from_args
    env_hetero.config_from_args = lambda *a, **k: orig(*a, **{**k, "arena_offset": meta["arena"]})
    try:
        env, args = _env_from_meta(meta, record_trace=True)
    finally:
        env_hetero.config_from_args = orig
    ep, checked = -1, 0
    for r in range(len(g["kind"])):
        if g["kind"][r] == 0:
            env.reset()
            ep += 1
            continue
        _, _, term, _, _ = env.step({1: g["actions"][r][0, :4].tolist(), 2: g["actions"][r][1, :3].tolist()})
        if term["__all__"]:
            rows, eps = env.world.trace_read()[0]
            rows = rows[eps == eps[-1]]                       # this episode: reset row + one row per tick
            want = ref[f"ep{ep}_aircraft"]
            assert len(ref[f"ep{ep}_rockets"]) == 0           # the reference never records rockets (add_unit does not)
            assert rows.shape[0] == want.shape[0], (ep, rows.shape, want.shape)
            for u in range(4):
                rec = ~np.isnan(want[:, u, 0])
                n = int(rec.sum())
                assert rec[:n].all(), "the reference records a unit without gaps until it is gone"
                assert np.allclose(rows[:n, u, :4], want[:n, u].astype(np.float32), rtol=0, atol=1e-5), (ep, u)
                alive = rows[:, u, 4] > 0
                # alive while recorded; the last recorded point may already carry alive = 0 (left the map in that tick)
                assert alive[: n - 1].all() and not alive[n:].any(), (ep, u, n)
                checked += n
    assert ep == 1 and checked > 500
    env.close()


def test_trace_ring_buffer_wraps_and_spans_episodes_in_hier():
    """hh_trace_enable on a HighLevelEnv world with auto-reset: rows of several episodes, ring wrap-around, rows = state history"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    w = World(make_config(n_arenas=40, env_kind=1, seed=2, auto_reset=True, horizon=40))
    w.trace_enable(3, capacity=64)
    w.reset()
    rng = np.random.default_rng(0)
    hist = [w.get_state()["ac_f"][:3, :, :2].astype(np.float32)]
    total = 1
    for step in range(9):
        cmd = torch.from_numpy(rng.integers(0, 3, (40, 3)).astype(np.int8)).cuda()
        tape = torch.from_numpy(np.stack([np.stack([rng.integers(0, 13, (40, 6)), rng.integers(0, 9, (40, 6)), rng.integers(0, 2, (40, 6)),
                                                    rng.integers(0, 2, (40, 6))], axis=-1) for _ in range(16)]).astype(np.int8)).cuda()
        w.hl_rollout(cmd, tape)
    tr = w.trace_read()
    assert len(tr) == 3
    for k, (rows, ep) in enumerate(tr):
        assert rows.shape[1:] == (6, 8) and len(rows) == 64          # more than 64 rows were written: the ring holds the last 64
        assert ep.max() >= 2 and (np.diff(ep) >= 0).all()             # several episodes, in order
        st = w.get_state()
        assert np.allclose(rows[-1, :, :2], st["ac_f"][k, :, :2].astype(np.float32))   # the newest row is the current state
    w.trace_enable(0, 0)                                              # off again
    w.hl_rollout(cmd, tape)


@pytest.mark.gpu
def test_highlevel_facade_batched_graph_path_equals_the_eager_loop():
    """HighLevelEnv with more than 64 arenas replays its commander step from a HIP graph (no early exit, NetPilot launches only):
    same observations, rewards, done flags and info as the same world stepped through macro_step eagerly"""
    import torch
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hier import HighLevelEnv, macro_step
    from hhmarl_2d_amd.world import World
    n = 200
    holder = {}

    class Lazy:   # HighLevelEnv wants its pilot at construction, the pilot wants the env's world
        def __call__(self, po, pm):
            return holder["pilot"](po, pm)
    env = HighLevelEnv({"args": make_args(1, horizon=60), "num_envs": n, "seed": 6, "pilot": Lazy()})
    env.pilot = holder["pilot"] = pilots.NetPilot(env.world, seed=2)            # the graph path needs the library's own NetPilot
    from hhmarl_2d_amd.env_hetero import config_from_args
    from hhmarl_2d_amd import _lib as L
    ref = World(config_from_args(env.args, L.ENV_HIGHLEVEL, n, 6))
    rp = pilots.NetPilot(ref, seed=2, bind=False)
    obs0, _ = env.reset()
    assert np.array_equal(obs0[1], ref.reset().cpu().numpy()[:, 0])
    rng = np.random.default_rng(3)
    dones = 0
    for t in range(12):
        cmd = {i: rng.integers(0, 3, n) for i in (1, 2, 3)}
        obs, rew, term, trunc, info = env.step(cmd)
        c = torch.from_numpy(np.stack([cmd[1], cmd[2], cmd[3]], axis=1).astype(np.int8)).cuda()
        o, r, v, d = [x.cpu().numpy() for x in macro_step(ref, c, rp)]
        for i in (1, 2, 3):
            assert np.array_equal(obs[i], o[:, i - 1]), (t, i)
            assert np.array_equal(rew[i], r[:, i - 1]), (t, i)
        assert np.array_equal(term["__all__"], d.astype(bool))
        dones += int(d.sum())
        fin = np.nonzero(d)[0]
        if len(fin):      # the facade leaves resets to the caller (RLlib's protocol): reset the same arenas on both sides
            m = np.zeros(n, dtype=np.uint8); m[fin] = 1
            env.world.reset(mask=torch.from_numpy(m).cuda()); ref.reset(mask=torch.from_numpy(m).cuda())
    assert dones > 0 and env._graph is not None
    env.close()
