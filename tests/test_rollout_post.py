"""Rollout post-processing next to the environment, CPU part: the centralised-critic rows against vectors
recorded from the reference's OWN callbacks (oracle/gen_critic_golden.py runs train_hetero.py:113-181 and
train_hier.py:100-165 behind a recording stand-in for RLlib's config builder) — exact equality; complete-episode slicing
against a plain loop."""
import os

import numpy as np
import torch

from hhmarl_2d_amd import rollout

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "critic_packing.npz")


def test_low_level_critic_rows_equal_the_reference_callback():
    g = np.load(GOLD)
    for mode in ("fight", "escape"):
        obs, act = torch.from_numpy(g[f"ll_{mode}_obs"]), torch.from_numpy(g[f"ll_{mode}_act"])
        for ag in (1, 2):
            rows = rollout.central_critic_rows(obs, act, ag)
            want = g[f"ll_{mode}_rows_agent{ag}"]
            assert rows.shape == want.shape and rows.dtype == torch.float32
            assert np.array_equal(rows.numpy(), want), (mode, ag)
        assert g[f"ll_{mode}_rows_agent1"].shape[1] == (57 if mode == "fight" else 66)


def test_commander_critic_rows_equal_the_reference_callback():
    g = np.load(GOLD)
    obs, act = torch.from_numpy(g["hl_obs"]), torch.from_numpy(g["hl_act"])
    for ag in (1, 2, 3):
        rows = rollout.central_critic_rows_hl(obs, act, ag)
        assert np.array_equal(rows.numpy(), g[f"hl_rows_agent{ag}"]), ag
    # batched leading axes work the same way: [T, N, ...]
    rows = rollout.central_critic_rows_hl(obs[:, None].expand(-1, 5, -1, -1), act[:, None].expand(-1, 5, -1), 2)
    assert rows.shape == (len(obs), 5, 105) and torch.equal(rows[:, 3], torch.from_numpy(g["hl_rows_agent2"]))


def test_complete_episode_slicing():
    rng = np.random.default_rng(0)
    done = (rng.random((200, 37)) < 0.03)
    seg, complete = rollout.episode_segments(torch.from_numpy(done))
    for n in range(37):
        k, last_done = 0, -1
        for t in range(200):
            assert int(seg[t, n]) == k
            if done[t, n]:
                k += 1
                last_done = t
        assert bool(complete[: last_done + 1, n].all()) and not bool(complete[last_done + 1:, n].any())
