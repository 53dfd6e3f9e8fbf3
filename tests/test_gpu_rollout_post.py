"""Rollout post-processing on the GPU: the GAE kernels against tests/golden/gae_vectors.npz — streams cut
from the committed reference traces with the advantages RLlib 2.4's postprocessing gives for them (oracle/gae_ref.py, restated
from ray==2.4.0 and pinned by hand-computed vectors in tests/test_rollout_post.py) — and against the same restatement on a real
rollout; the centralised-critic rows on device tensors against the reference's own callbacks' output."""
import os

import numpy as np
import pytest

from helpers import random_actions

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gae_kernels_reproduce_the_committed_vectors():
    import torch
    from hhmarl_2d_amd.rollout import gae, gae_rllib
    g = np.load(os.path.join(GOLDEN, "gae_vectors.npz"))
    tags = sorted({k.split("/")[0] for k in g.files})
    assert len(tags) == 4
    for tag in tags:
        gamma, lam = g[f"{tag}/gamma_lambda"]
        r, v = torch.from_numpy(g[f"{tag}/reward"]).cuda(), torch.from_numpy(g[f"{tag}/value"]).cuda()
        valid, done = torch.from_numpy(g[f"{tag}/valid"]).cuda(), torch.from_numpy(g[f"{tag}/done"]).cuda()
        assert float((r * (valid == 0)).abs().sum()) == 0.0          # the world (and the reference) report 0 where there is no key
        adv, ret = gae_rllib(r, v, done, float(gamma), float(lam))
        assert np.array_equal(adv.cpu().numpy(), g[f"{tag}/adv_rllib"]), f"{tag}: advantages (RLlib semantics) differ from the restatement"
        assert np.array_equal(ret.cpu().numpy(), g[f"{tag}/ret_rllib"]), f"{tag}: value targets (RLlib semantics)"
        adv_m, ret_m = gae(r, v, valid, done, float(gamma), float(lam))
        assert np.abs(adv_m.cpu().numpy() - g[f"{tag}/adv_masked"]).max() <= 1e-5 and np.abs(ret_m.cpu().numpy() - g[f"{tag}/ret_masked"]).max() <= 1e-5, tag
    # the two conventions do differ where an agent died inside an episode
    tag = "l3_fight_pursuit_share"
    assert np.abs(g[f"{tag}/adv_rllib"] - g[f"{tag}/adv_masked"]).max() > 1e-2


def test_gae_kernels_on_a_real_rollout():
    import torch
    import gae_ref
    from hhmarl_2d_amd.rollout import central_critic_inputs, gae, gae_rllib
    from hhmarl_2d_amd.world import World, make_config
    N, T = 700, 150
    w = World(make_config(n_arenas=N, level=3, seed=3, auto_reset=True, horizon=60))
    w.reset()
    act = torch.from_numpy(random_actions(np.random.default_rng(0), (T, N), 2)).cuda()
    obs, rew, val, done = w.rollout(act)
    assert done.any() and (val == 0).any() and float((rew * (val == 0)).abs().sum()) == 0.0
    value = torch.randn((T + 1, N, 2), device="cuda")
    h = [x.cpu().numpy() for x in (rew, val, value, done)]
    for lam in (0.95, 1.0):
        adv, ret = gae_rllib(rew, value, done, 0.99, lam)
        a_ref, r_ref = gae_ref.rllib_stream(h[0], h[1], h[2], h[3], 0.99, lam)
        assert np.array_equal(adv.cpu().numpy(), a_ref) and np.array_equal(ret.cpu().numpy(), r_ref), lam
    adv, ret = gae(rew, value, val, done, 0.99, 0.95)
    a_ref, r_ref = gae_ref.masked_stream(h[0], h[1], h[2], h[3], 0.99, 0.95)
    assert np.abs(adv.cpu().numpy() - a_ref).max() <= 1e-4 and np.abs(ret.cpu().numpy() - r_ref).max() <= 1e-4
    cc = central_critic_inputs(obs, act)
    assert cc[1]["obs_1_own"].shape == (T, N, 26) and cc[1]["obs_2"].shape == (T, N, 24)
    assert cc[2]["act_1_own"].shape == (T, N, 3) and cc[2]["act_2"].shape == (T, N, 4)
    assert float(cc[1]["act_1_own"][..., 0].max()) <= 1.0 and float(cc[1]["act_1_own"][..., 1].max()) <= 1.0


def test_critic_rows_on_device_tensors_equal_the_reference_callbacks():
    """central_critic_rows / central_critic_rows_hl on CUDA tensors against tests/golden/critic_packing.npz (recorded from
    train_hetero.py:113-181 and train_hier.py:100-165 themselves): exact equality, as on the CPU"""
    import torch
    from hhmarl_2d_amd import rollout
    g = np.load(os.path.join(GOLDEN, "critic_packing.npz"))
    for mode in ("fight", "escape"):
        obs, act = torch.from_numpy(g[f"ll_{mode}_obs"]).cuda(), torch.from_numpy(g[f"ll_{mode}_act"]).cuda()
        for ag in (1, 2):
            rows = rollout.central_critic_rows(obs, act, ag)
            assert rows.is_cuda and np.array_equal(rows.cpu().numpy(), g[f"ll_{mode}_rows_agent{ag}"]), (mode, ag)
    obs, act = torch.from_numpy(g["hl_obs"]).cuda(), torch.from_numpy(g["hl_act"]).cuda()
    for ag in (1, 2, 3):
        assert np.array_equal(rollout.central_critic_rows_hl(obs, act, ag).cpu().numpy(), g[f"hl_rows_agent{ag}"]), ag
