"""-m gpu: 64 sub-environments driven through the BaseEnv surface reproduce 64 single-arena LowLevelEnv facades (each created with its own
arena_offset), observation for observation, reward key for reward key, done for done, through episode ends and resets."""
import numpy as np
import pytest

from test_vector_env import make_args, sample_actions


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fight", "escape"])
def test_vector_env_equals_single_env_facades(mode):
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    from hhmarl_2d_amd.vector_env import LowLevelVectorEnv
    N, args = 64, make_args(level=3, mode=mode, horizon=30)
    venv = LowLevelVectorEnv({"args": args, "num_envs": N, "seed": 5, "arena_offset": 1000})
    singles = [LowLevelEnv({"args": args, "seed": 5, "arena_offset": 1000 + i}) for i in range(N)]
    d1, d2 = (26, 24) if mode == "fight" else (30, 29)
    obs, rew, term, trunc, info, _ = venv.poll()
    for i, s in enumerate(singles):
        o, _ = s.reset()
        assert np.array_equal(o[1], obs[i][1]) and np.array_equal(o[2], obs[i][2]) and obs[i][1].shape == (d1,) and obs[i][2].shape == (d2,)
    rng = np.random.default_rng(1)
    dones = 0
    for it in range(80):
        acts = sample_actions(rng, range(N))
        venv.send_actions(acts)
        obs, rew, term, trunc, info, _ = venv.poll()
        assert sorted(obs) == list(range(N))
        for i, s in enumerate(singles):
            o, r, t, tr, inf = s.step(acts[i])
            assert np.array_equal(o[1], obs[i][1]) and np.array_equal(o[2], obs[i][2])
            assert r == rew[i] and t == term[i] and tr == trunc[i] and inf == info[i] == {}
            if t["__all__"]:
                dones += 1
                ro, ri = venv.try_reset(i)
                so, _ = s.reset()
                assert np.array_equal(so[1], ro[i][1]) and np.array_equal(so[2], ro[i][2]) and ri == {i: {}}
    assert dones >= 2 * N
    venv.stop()
    for s in singles:
        s.close()


class _LazyNets:
    """`opponent_policy` callable for levels 4-5: a random-init bank of the reference architectures (the same weights for every seed-equal instance),
    built inside its first call — on whatever owns the world: a LowLevelEnv facade or the vector adapter's backend"""

    def __init__(self):
        self.nets = None

    def __call__(self, opp_obs, env):
        from hhmarl_2d_amd import pilots
        if self.nets is None:
            self.nets = pilots.OpponentNets(env.world, seed=3)
        return self.nets(opp_obs, env)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [4, 5])
def test_vector_env_levels_4_and_5_equal_single_env_facades(level):
    """levels 4-5: the opponents fly frozen policies between the two halves of the step (env_hetero.py:160-172; level 5 draws the policy set per arena and
    episode, env_hetero.py:55-59): 32 sub-environments behind the BaseEnv surface against 32 single-arena facades, through episode ends and resets"""
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    from hhmarl_2d_amd.vector_env import LowLevelVectorEnv
    N, args = 32, make_args(level=level, mode="fight", horizon=25)
    with pytest.raises(ValueError):
        LowLevelVectorEnv({"args": args, "num_envs": N})
    venv = LowLevelVectorEnv({"args": args, "num_envs": N, "seed": 9, "arena_offset": 300, "opponent_policy": _LazyNets()})
    singles = [LowLevelEnv({"args": args, "seed": 9, "arena_offset": 300 + i, "opponent_policy": _LazyNets()}) for i in range(N)]
    obs, rew, term, trunc, info, _ = venv.poll()
    for i, s in enumerate(singles):
        o, _ = s.reset()
        assert np.array_equal(o[1], obs[i][1]) and np.array_equal(o[2], obs[i][2])
    if level == 5:
        assert set(np.unique(venv.b.opp_k)) <= {3, 4, 5} and [s.opp_k for s in singles] == list(venv.b.opp_k)
    rng = np.random.default_rng(2)
    dones, ks = 0, set()
    for it in range(70):
        acts = sample_actions(rng, range(N))
        venv.send_actions(acts)
        obs, rew, term, trunc, info, _ = venv.poll()
        for i, s in enumerate(singles):
            o, r, t, tr, inf = s.step(acts[i])
            assert np.array_equal(o[1], obs[i][1]) and np.array_equal(o[2], obs[i][2]), f"step {it}, sub-environment {i}"
            assert r == rew[i] and t == term[i] and tr == trunc[i]
            if t["__all__"]:
                dones += 1
                ro, ri = venv.try_reset(i)
                so, _ = s.reset()
                assert np.array_equal(so[1], ro[i][1]) and np.array_equal(so[2], ro[i][2])
                if level == 5:
                    assert s.opp_k == venv.b.opp_k[i]
                    ks.add(int(s.opp_k))
    assert dones >= N
    if level == 5:
        assert ks == {3, 4, 5}, "the episodes must cover the three policy sets"
    venv.stop()
    for s in singles:
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["sides", "variants", "4v5"])
@pytest.mark.parametrize("N,iters", [(16, 40), (70, 25)], ids=["16-eager-early-exit", "70-hip-graph"])
def test_highlevel_vector_env_equals_single_env_facades(N, iters, form, monkeypatch):
    """HighLevelVectorEnv (one commander step of every sub-environment per send_actions: eager with the early exit for a handful, one replayed HIP graph
    above 64) against N single-arena HighLevelEnv facades with the same random-init pilot networks, through episode ends, masked resets and the eval counters"""
    from test_vector_env import make_hl_args
    from hhmarl_2d_amd import _lib as L
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    from hhmarl_2d_amd.pilots import NetPilot, VariantNetPilot
    from hhmarl_2d_amd.vector_env import HighLevelVectorEnv
    # form: "sides" = two pilot calls per sub-step on both sides of the comparison; "variants" = the vector env flies the variant-row pilot (what the facade
    # builds itself from a policy_dir: one launch + one policy call per sub-step) against single environments on the two-call path; "4v5" = ten-slot arenas
    if form == "variants":   # both pilots on the tile form of one width: a row's logits must not depend on which call evaluates it
        monkeypatch.setenv("HH_POLICY_W", "0")
        monkeypatch.setenv("HH_POLICY_TILE", "32")
    nA, nO = (4, 5) if form == "4v5" else (3, 3)
    ids = tuple(range(1, nA + 1))
    args = make_hl_args(horizon=45, eval_info=True, num_agents=nA, num_opps=nO)
    placeholder = lambda po, pm: None
    venv = HighLevelVectorEnv({"args": args, "num_envs": N, "seed": 4, "arena_offset": 700, "pilot": placeholder})
    if form == "variants":
        venv.b.pilot = VariantNetPilot(venv.b.world, seed=9)
        venv.b._pbuf = venv.b.world.alloc_pilot_variants()
    else:
        venv.b.pilot = NetPilot(venv.b.world, seed=9)
    if N == 16:
        venv.b._graph_from = 1 << 30   # this case keeps the eager path with the early exit covered (the default replays a HIP graph at every size)
    singles = []
    for i in range(N):
        s = HighLevelEnv({"args": args, "seed": 4, "arena_offset": 700 + i, "pilot": placeholder})
        s.pilot = NetPilot(s.world, seed=9)
        singles.append(s)
    obs, rew, term, trunc, info, _ = venv.poll()
    for i, s in enumerate(singles):
        o, _ = s.reset()
        assert all(np.array_equal(o[k], obs[i][k]) for k in ids) and obs[i][1].shape == (34,)
    rng = np.random.default_rng(6)
    dones = 0
    for it in range(iters):
        acts = {e: {k: int(rng.integers(3)) for k in ids} for e in range(N)}
        venv.send_actions(acts)
        obs, rew, term, trunc, info, _ = venv.poll()
        for i, s in enumerate(singles):
            o, r, t, tr, inf = s.step(acts[i])
            assert all(np.array_equal(o[k], obs[i][k]) for k in ids), f"step {it}, sub-environment {i}"
            assert r == rew[i] and t == term[i] and tr == trunc[i] and inf == info[i] and set(inf) == set(L.EVAL_KEYS)
            if t["__all__"]:
                dones += 1
                ro, ri = venv.try_reset(i)
                so, _ = s.reset()
                assert all(np.array_equal(so[k], ro[i][k]) for k in ids)
    assert dones >= N // 4
    venv.stop()
    for s in singles:
        s.close()
