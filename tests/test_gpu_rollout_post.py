"""GAE kernel against a float32 NumPy reference of the same recursion, on a real rollout."""
import numpy as np
import pytest

from helpers import random_actions

pytestmark = pytest.mark.gpu


def _gae_ref(r, v, valid, done, gamma, lam):
    T = r.shape[0]
    adv = np.zeros_like(r)
    ret = np.zeros_like(r)
    a_next = np.zeros(r.shape[1:], dtype=np.float32)
    v_next = v[T]
    for t in range(T - 1, -1, -1):
        nd = (1.0 - done[t].astype(np.float32))[:, None]
        delta = r[t] + np.float32(gamma) * v_next * nd - v[t]
        a = delta + np.float32(gamma) * np.float32(lam) * nd * a_next
        a = np.where(valid[t] > 0, a, 0).astype(np.float32)
        adv[t] = a
        ret[t] = np.where(valid[t] > 0, a + v[t], 0)
        a_next = a
        v_next = v[t]
    return adv, ret


def test_gae_matches_numpy_on_a_rollout():
    import torch
    from hhmarl_2d_amd.rollout import central_critic_inputs, gae
    from hhmarl_2d_amd.world import World, make_config
    N, T = 1000, 120
    w = World(make_config(n_arenas=N, level=3, seed=3, auto_reset=True, horizon=60))
    w.reset()
    act = torch.from_numpy(random_actions(np.random.default_rng(0), (T, N), 2)).cuda()
    obs, rew, val, done = w.rollout(act)
    assert done.any() and (val == 0).any()
    value = torch.randn((T + 1, N, 2), device="cuda")
    adv, ret = gae(rew, value, val, done, 0.99, 0.95)
    a_ref, r_ref = _gae_ref(rew.cpu().numpy(), value.cpu().numpy(), val.cpu().numpy(), done.cpu().numpy(), 0.99, 0.95)
    assert np.abs(adv.cpu().numpy() - a_ref).max() <= 1e-4 and np.abs(ret.cpu().numpy() - r_ref).max() <= 1e-4
    cc = central_critic_inputs(obs, act)
    assert cc[1]["obs_1_own"].shape == (T, N, 26) and cc[1]["obs_2"].shape == (T, N, 24)
    assert cc[2]["act_1_own"].shape == (T, N, 3) and cc[2]["act_2"].shape == (T, N, 4)
    assert float(cc[1]["act_1_own"][..., 0].max()) <= 1.0 and float(cc[1]["act_1_own"][..., 1].max()) <= 1.0
