"""The boundary under a REAL RLlib: `ray` is not installed in the build image or on the GPU box, so every other test
drives the facades through a scripted stand-in of RLlib 2.4's call order (tests/test_vector_env.py).  This file is the guard for the
day the package exists: it skips with a printed reason today and, once `import ray` works, builds the trainers' configurations the way
the reference does (train_hetero.py:210-245 `PPOConfig().rollouts(...).environment(env=LowLevelEnv, env_config=args.env_config)
.multi_agent(..., observation_fn=central_critic_observer)`, train_hier.py:180-206) around the drop-in classes and runs ONE
`algo.train()` each — with RLlib's default fully connected model on the observer's Dict space (the reference's Fight1/Fight2 torch
modules are trainer code, out of scope: SURVEY.md 2 rows 10-16).

The class-level checks need no GPU; the `algo.train()` runs are `-m gpu` (the facades raise without the HIP library and a device)."""
import numpy as np
import pytest

ray = pytest.importorskip("ray", reason="ray[rllib] is not installed in this image: the RLlib smoke test fires the day it is")
pytest.importorskip("ray.rllib", reason="ray is installed without rllib")

from hhmarl_2d_amd.config import make_args  # noqa: E402


def _spaces():
    try:
        from gymnasium import spaces
    except ImportError:   # older ray
        from gym import spaces
    return spaces


def _observer_2v2(fight=True):
    """central_critic_observer and its spaces as train_hetero.py:162-198 defines them (own obs, the other agent's obs, both action slots zero
    while sampling: on_postprocess_trajectory fills them afterwards)"""
    spaces = _spaces()
    d1, d2 = (26, 24) if fight else (30, 29)

    def observer(agent_obs, **kw):
        return {1: {"obs_1_own": agent_obs[1], "obs_2": agent_obs[2], "act_1_own": np.zeros(4), "act_2": np.zeros(3)},
                2: {"obs_1_own": agent_obs[2], "obs_2": agent_obs[1], "act_1_own": np.zeros(3), "act_2": np.zeros(4)}}

    def space(own, oth, a_own, a_oth):
        return spaces.Dict({"obs_1_own": spaces.Box(low=0, high=1, shape=(own,)), "obs_2": spaces.Box(low=0, high=1, shape=(oth,)),
                            "act_1_own": spaces.Box(low=0, high=12, shape=(a_own,), dtype=np.float32),
                            "act_2": spaces.Box(low=0, high=12, shape=(a_oth,), dtype=np.float32)})
    return observer, space(d1, d2, 4, 3), space(d2, d1, 3, 4)


def _ppo_2v2(env, env_config):
    from ray.rllib.algorithms.ppo import PPOConfig
    from ray.rllib.policy.policy import PolicySpec
    spaces = _spaces()
    observer, sp1, sp2 = _observer_2v2()
    return (PPOConfig()
            .rollouts(num_rollout_workers=0, batch_mode="complete_episodes", enable_connectors=False)     # train_hetero.py:212
            .resources(num_gpus=0)
            .evaluation(evaluation_interval=None)
            .environment(env=env, env_config=env_config, disable_env_checking=True)                        # train_hetero.py:215
            .training(kl_target=0.025, train_batch_size=600, gamma=0.99, clip_param=0.25, lr=1e-4, lambda_=0.95, sgd_minibatch_size=200, num_sgd_iter=1)
            .framework("torch")
            .multi_agent(policies={"ac1_policy": PolicySpec(None, sp1, spaces.MultiDiscrete([13, 9, 2, 2]), {}),
                                   "ac2_policy": PolicySpec(None, sp2, spaces.MultiDiscrete([13, 9, 2]), {})},
                         policy_mapping_fn=lambda agent_id, episode, worker, **kw: f"ac{agent_id}_policy",
                         policies_to_train=["ac1_policy", "ac2_policy"], observation_fn=observer))


def test_facades_are_rllib_classes():
    """with ray importable the drop-ins ARE RLlib environments (env_hetero.py / vector_env.py pick the base class at import)"""
    from ray.rllib.env.base_env import BaseEnv
    from ray.rllib.env.multi_agent_env import MultiAgentEnv
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    from hhmarl_2d_amd.vector_env import HighLevelVectorEnv, LowLevelVectorEnv
    assert issubclass(LowLevelEnv, MultiAgentEnv) and issubclass(HighLevelEnv, MultiAgentEnv)
    assert issubclass(LowLevelVectorEnv, BaseEnv) and issubclass(HighLevelVectorEnv, BaseEnv)


@pytest.mark.gpu
def test_ppo_trains_one_iteration_on_lowlevel_env():
    """train_hetero.py:210-245 with the import swapped: one algo.train() on LowLevelEnv, level 1"""
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    args = make_args(0, level=1)
    ray.init(ignore_reinit_error=True, num_cpus=2, include_dashboard=False)
    try:
        algo = _ppo_2v2(LowLevelEnv, args.env_config).build()
        r = algo.train()
        assert r["timesteps_total"] > 0 and np.isfinite(r["episode_reward_mean"])
        algo.stop()
    finally:
        ray.shutdown()


@pytest.mark.gpu
def test_ppo_trains_one_iteration_on_the_vector_env():
    """N arenas on the GPU behind RLlib's sampler through the BaseEnv surface"""
    from ray import tune
    from hhmarl_2d_amd.vector_env import LowLevelVectorEnv
    args = make_args(0, level=1)
    tune.register_env("hh_vector", lambda cfg: LowLevelVectorEnv(cfg))
    ray.init(ignore_reinit_error=True, num_cpus=2, include_dashboard=False)
    try:
        algo = _ppo_2v2("hh_vector", dict(args.env_config, num_envs=64, seed=3)).build()
        r = algo.train()
        assert r["timesteps_total"] > 0
        algo.stop()
    finally:
        ray.shutdown()


@pytest.mark.gpu
def test_ppo_trains_one_iteration_on_highlevel_env():
    """train_hier.py:180-206: the commander policy on HighLevelEnv (3-vs-3), pilots = the scripted fallback tape of the facade's own default"""
    from ray.rllib.algorithms.ppo import PPOConfig
    from ray.rllib.policy.policy import PolicySpec
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    spaces = _spaces()
    args = make_args(1)

    def observer(agent_obs, **kw):   # train_hier.py:150-168
        ids = (1, 2, 3)
        out = {}
        for i in ids:
            o = [j for j in ids if j != i]
            out[i] = {"obs_1_own": agent_obs[i], "obs_2": agent_obs[o[0]], "obs_3": agent_obs[o[1]],
                      "act_1_own": np.zeros(1), "act_2": np.zeros(1), "act_3": np.zeros(1)}
        return out
    box = lambda n, hi=1: spaces.Box(low=0, high=hi, shape=(n,), dtype=np.float32)   # noqa: E731
    sp = spaces.Dict({"obs_1_own": box(34), "obs_2": box(34), "obs_3": box(34), "act_1_own": box(1, 2), "act_2": box(1, 2), "act_3": box(1, 2)})
    import torch

    def pilot(pilot_obs, pilot_mode):   # stands where the reference torch.load()s its frozen fight / escape policies (env_base.py:312-347)
        n, dev = pilot_obs.shape[0], pilot_obs.device
        hi = torch.tensor([13, 9, 2, 2], device=dev)
        return (torch.rand((n, 6, 4), device=dev) * hi).to(torch.int8)
    ray.init(ignore_reinit_error=True, num_cpus=2, include_dashboard=False)
    try:
        algo = (PPOConfig()
                .rollouts(num_rollout_workers=0, batch_mode="complete_episodes", enable_connectors=False)
                .resources(num_gpus=0)
                .environment(env=HighLevelEnv, env_config=dict(args.env_config, pilot=pilot), disable_env_checking=True)
                .training(train_batch_size=200, sgd_minibatch_size=100, num_sgd_iter=1, kl_target=0.05, gamma=0.99, clip_param=0.25, lr=1e-4)
                .framework("torch")
                .multi_agent(policies={"commander_policy": PolicySpec(None, sp, spaces.Discrete(3), {})},
                             policy_mapping_fn=lambda agent_id, episode, worker, **kw: "commander_policy", observation_fn=observer)
                .build())
        r = algo.train()
        assert r["timesteps_total"] > 0
        algo.stop()
    finally:
        ray.shutdown()
