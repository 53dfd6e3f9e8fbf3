"""The RLlib-shaped vector surface (hhmarl_2d_amd/vector_env.py: BaseEnv protocol of ray/rllib/env/base_env.py, RLlib 2.4).  CPU part: the
protocol logic against a scripted stand-in of the call sequence RLlib's env runner makes (sampler.py `_env_runner`: poll -> try_reset
of every sub-environment that reports "__all__" -> send_actions for everything polled), with the CPU oracle as the world behind it, checked
sub-environment by sub-environment against N independent single-arena oracle worlds.  GPU part: tests/test_gpu_vector_env.py."""
import types

import numpy as np
import pytest

from hhmarl_2d_amd import _lib as L
from hhmarl_2d_amd.env_hetero import config_from_args
from hhmarl_2d_amd.vector_env import LowLevelVectorEnv


def make_args(level=3, mode="fight", horizon=40):
    return types.SimpleNamespace(level=level, agent_mode=mode, num_agents=2, num_opps=2, horizon=horizon, friendly_kill=True, friendly_punish=False,
                                 esc_dist_rew=False, map_size=0.3, glob_frac=0.0, rew_scale=1)


class OracleBackend:
    """the reset / step surface of vector_env._GpuBackend on the CPU oracle (test infrastructure)"""

    def __init__(self, oracle, cfg):
        self.w = oracle.OracleWorld(cfg)
        self.N, self.n_agents, self.D = self.w.N, self.w.n_agents, self.w.D
        self.act_host = np.zeros((self.N, self.w.n_ctrl, 4), dtype=np.int8)
        self.mask_host = np.zeros((self.N,), dtype=np.uint8)
        self.calls = []

    def reset(self, masked):
        self.calls.append(("reset", int(self.mask_host.sum()) if masked else self.N))
        return self.w.reset(self.mask_host.copy() if masked else None)

    def step(self):
        self.calls.append(("step", self.N))
        return self.w.step(self.act_host.copy())

    def close(self):
        pass


def sample_actions(rng, env_ids):
    return {e: {1: np.array([rng.integers(13), rng.integers(9), rng.integers(2), rng.integers(2)]),
                2: np.array([rng.integers(13), rng.integers(9), rng.integers(2)])} for e in env_ids}


def runner_loop(env, rng, iters, on_step=None):
    """the order of calls of RLlib 2.4's `_env_runner` (ray/rllib/evaluation/sampler.py): poll; for every sub-environment whose
    terminateds / truncateds carry "__all__": try_reset and continue with the returned observation; send_actions for every
    sub-environment that has an observation.  Returns per sub-environment the list of (obs, rewards, done) it saw, resets marked."""
    seen = {}
    for _ in range(iters):
        obs, rew, term, trunc, info, off = env.poll()
        assert off == {} and set(obs) == set(rew) == set(term) == set(trunc) == set(info)
        to_act = []
        for e in sorted(obs):
            assert term[e] is trunc[e] or term[e] == trunc[e]
            seen.setdefault(e, []).append(("step", obs[e], rew[e], term[e]["__all__"]))
            if term[e]["__all__"] or trunc[e]["__all__"]:
                ro, ri = env.try_reset(e)
                assert set(ro) == {e} and ri == {e: {}}
                seen[e].append(("reset", ro[e], {}, False))
            to_act.append(e)
        acts = sample_actions(rng, to_act)
        if on_step:
            on_step(acts)
        env.send_actions(acts)
    return seen


def test_protocol_against_independent_single_arena_worlds(oracle):
    N, args = 12, make_args(horizon=25)
    mk = lambda n, off: config_from_args(args, L.ENV_LOWLEVEL, n, 7, auto_reset=False, arena_offset=off)
    cfg = mk(N, 100)
    env = LowLevelVectorEnv({"args": args, "num_envs": N, "_backend": OracleBackend(oracle, cfg)})
    assert env.get_agent_ids() == {1, 2} and env.get_sub_environments() == [] and env.observation_space[1].shape == (26,)
    singles = [oracle.OracleWorld(mk(1, 100 + i)) for i in range(N)]
    first = [w.reset() for w in singles]
    log = []
    seen = runner_loop(env, np.random.default_rng(3), 70, on_step=log.append)
    # replay every sub-environment's own action stream on its own single-arena world, resetting where the vector env reported done
    n_done = 0
    for e in range(N):
        w, it = singles[e], iter(seen[e])
        kind, o, r, d = next(it)
        assert kind == "step" and r == {} and not d and np.array_equal(o[1], first[e][0, 0, :26]) and np.array_equal(o[2], first[e][0, 1, :24])
        for acts in log[:-1]:
            a = np.zeros((1, 2, 4), dtype=np.int8)
            a[0, 0] = acts[e][1]
            a[0, 1, :3] = acts[e][2]
            wo, wr, wv, wd = w.step(a)
            kind, o, r, d = next(it)
            assert kind == "step" and d == bool(wd[0])
            assert np.array_equal(o[1], wo[0, 0, :26]) and np.array_equal(o[2], wo[0, 1, :24]) and o[1].dtype == np.float32
            assert r == {i: float(wr[0, i - 1]) for i in (1, 2) if wv[0, i - 1]}          # rewards only for ids alive at step start
            if d:
                n_done += 1
                ro = w.reset()
                kind, o, r, d2 = next(it)
                assert kind == "reset" and np.array_equal(o[1], ro[0, 0, :26]) and np.array_equal(o[2], ro[0, 1, :24])
    assert n_done >= N   # horizon 25 in 70 iterations: every sub-environment finished at least twice... at least once each on average
    # one device step per iteration and at most one masked reset behind it: never a call per sub-environment
    calls = env.b.calls
    assert sum(1 for c in calls if c[0] == "step") == 70 and sum(1 for c in calls if c[0] == "reset") <= 71


def test_protocol_errors_and_corner_cases(oracle):
    N, args = 4, make_args(horizon=3)
    env = LowLevelVectorEnv({"args": args, "num_envs": N, "_backend": OracleBackend(oracle, config_from_args(args, L.ENV_LOWLEVEL, N, 1))})
    obs, *_ = env.poll()
    assert sorted(obs) == [0, 1, 2, 3]
    assert env.poll()[0] == {}                                     # nothing new until actions are sent
    rng = np.random.default_rng(0)
    with pytest.raises(ValueError, match="no action"):
        env.send_actions(sample_actions(rng, [0, 1, 2]))           # a running sub-environment without an action
    for _ in range(3):
        env.send_actions(sample_actions(rng, range(N)))
        obs, rew, term, trunc, info, _ = env.poll()
    assert all(term[e]["__all__"] for e in range(N))               # horizon 3
    with pytest.raises(ValueError, match="already done"):
        env.send_actions(sample_actions(rng, range(N)))
    o0, i0 = env.try_reset(0)
    assert set(o0[0]) == {1, 2} and i0 == {0: {}}
    o0b, _ = env.try_reset(0)                                      # reset twice without stepping: a new episode again (like env.reset() twice)
    assert set(o0b[0]) == {1, 2} and not np.array_equal(o0[0][1], o0b[0][1])
    for e in (1, 2, 3):
        env.try_reset(e)
    env.send_actions(sample_actions(rng, range(N)))
    obs, rew, term, *_ = env.poll()
    assert sorted(obs) == [0, 1, 2, 3] and not any(term[e]["__all__"] for e in range(N))
    ro, _ = env.try_reset(2)                                       # a reset in the middle of an episode
    assert set(ro) == {2}
    env.send_actions(sample_actions(rng, range(N)))
    assert sorted(env.poll()[0]) == [0, 1, 2, 3]
    with pytest.raises(ValueError, match="levels 4-5"):
        LowLevelVectorEnv({"args": make_args(level=4), "num_envs": 2})
    with pytest.raises(ValueError, match="env_config\\['args'\\] is missing"):
        LowLevelVectorEnv({"num_envs": 2})                           # a clear error instead of an AttributeError on None


# ------------------------------------------------------------------------------------------------ HighLevelVectorEnv
def make_hl_args(horizon=60, eval_info=False, num_agents=3, num_opps=3):
    return types.SimpleNamespace(level=5, agent_mode="fight", num_agents=num_agents, num_opps=num_opps, horizon=horizon, friendly_kill=True, friendly_punish=False,
                                 esc_dist_rew=False, map_size=0.5, glob_frac=0.0, rew_scale=1, hier_action_assess=True, hier_opp_fight_ratio=75,
                                 eval_info=eval_info, eval_hl=True)


def row_pilot(pilot_obs, pilot_mode):
    """a 'frozen pilot' that is a function of the unit's own observation row only, so that it flies a unit the same way in any batch"""
    h = np.floor(np.abs(pilot_obs.astype(np.float64) @ np.linspace(0.37, 3.1, pilot_obs.shape[-1])) * 9973.0).astype(np.int64)
    a = np.stack([h % 13, (h // 13) % 9, (h // 117) % 2, (h // 234) % 2], axis=-1).astype(np.int8)
    a[pilot_mode == 0] = 0
    return a


def oracle_macro_step(w, cmd):
    """env_hier.py:114-140 on the CPU oracle with row_pilot between the phases"""
    nA = w.n_agents
    w.hl_begin(cmd)
    for sub in range(16):
        act = row_pilot(*w.hl_pilot_obs(0))
        w.hl_agents_act(act)
        act[:, nA:] = row_pilot(*w.hl_pilot_obs(1))[:, nA:]
        if w.hl_tick(act) == 0:
            break
    return w.hl_end()


class OracleHierBackend:
    """the reset / step surface of vector_env._GpuHierBackend on the CPU oracle (test infrastructure)"""

    def __init__(self, oracle, cfg):
        self.w = oracle.OracleWorld(cfg)
        self.N, self.n_agents, self.D = self.w.N, self.w.n_agents, self.w.D
        self.act_host = np.zeros((self.N, self.n_agents), dtype=np.int8)
        self.mask_host = np.zeros((self.N,), dtype=np.uint8)

    def reset(self, masked):
        return self.w.reset(self.mask_host.copy() if masked else None)

    def step(self):
        return oracle_macro_step(self.w, self.act_host.copy())

    def eval_info(self):
        return self.w.eval_info()[0]

    def close(self):
        pass


def test_highlevel_protocol_against_independent_single_arena_worlds(oracle):
    """HighLevelVectorEnv: N commander environments behind the BaseEnv surface = N single-arena HighLevelEnv worlds, commander step for commander step
    (observations of agents 1..3, rewards only for the ids alive at step start, done, the eval counters in the info dict), through episode ends and resets"""
    from hhmarl_2d_amd.vector_env import HighLevelVectorEnv
    N, args = 6, make_hl_args(horizon=40, eval_info=True)
    mk = lambda n, off: config_from_args(args, L.ENV_HIGHLEVEL, n, 11, auto_reset=False, arena_offset=off)
    with pytest.raises(ValueError, match="frozen low-level pilot"):
        HighLevelVectorEnv({"args": args, "num_envs": N})
    env = HighLevelVectorEnv({"args": args, "num_envs": N, "_backend": OracleHierBackend(oracle, mk(N, 40))})
    assert env.get_agent_ids() == {1, 2, 3} and env.observation_space.shape == (34,) and env.action_space.n == 3
    singles = [oracle.OracleWorld(mk(1, 40 + i)) for i in range(N)]
    obs, rew, term, trunc, info, _ = env.poll()
    for e, w in enumerate(singles):
        ro = w.reset()
        assert all(np.array_equal(obs[e][i], ro[0, i - 1]) for i in (1, 2, 3)) and obs[e][1].shape == (34,) and rew[e] == {} and info[e] == {}
    rng = np.random.default_rng(5)
    n_done = 0
    for it in range(40):
        acts = {e: {i: int(rng.integers(3)) for i in (1, 2, 3)} for e in range(N)}
        env.send_actions(acts)
        obs, rew, term, trunc, info, _ = env.poll()
        assert sorted(obs) == list(range(N))
        for e, w in enumerate(singles):
            wo, wr, wv, wd = oracle_macro_step(w, np.array([[acts[e][1], acts[e][2], acts[e][3]]], dtype=np.int8))
            assert all(np.array_equal(obs[e][i], wo[0, i - 1]) for i in (1, 2, 3)), f"step {it}, sub-environment {e}"
            assert rew[e] == {i: float(wr[0, i - 1]) for i in (1, 2, 3) if wv[0, i - 1]} and term[e] == trunc[e] == {"__all__": bool(wd[0])}
            assert info[e] == {k: int(w.eval_info()[0][0, j]) for j, k in enumerate(L.EVAL_KEYS)}
            if wd[0]:
                n_done += 1
                ro, ri = env.try_reset(e)
                so = w.reset()
                assert all(np.array_equal(ro[e][i], so[0, i - 1]) for i in (1, 2, 3)) and ri == {e: {}}
    assert n_done >= 3
