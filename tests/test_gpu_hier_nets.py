"""hh_hl_step_nets: HighLevelEnv.step with the pilot networks inside ONE cooperative launch (hh_kernels_coop.h) against the
launch-by-launch path (hh_hl_begin, 16 x { policy, hh_hl_agents_act, policy, hh_hl_tick }, hh_hl_end = env_hier.macro_step with
a NetPilot): commander observations, rewards, reward keys, done flags, eval counters, event masks and the whole final state
bit for bit, over several commander steps with auto-reset, on a full grid, a ragged small world and an n-vs-m world."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,nA,nO,steps", [(8192, 3, 3, 5), (203, 3, 3, 7), (1, 3, 3, 8), (640, 2, 3, 6), (3000, 3, 1, 6)],
                         ids=["8192-3v3", "203-3v3", "1-3v3", "640-2v3", "3000-3v1"])
def test_cooperative_step_equals_the_launch_by_launch_path(N, nA, nO, steps, monkeypatch):
    import torch
    monkeypatch.setenv("HH_POLICY_W", "0")   # the cooperative step walks tile-form tiles: bit equality is against that form of the separate launches
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    kw = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=77, auto_reset=True, horizon=40, arena_offset=5)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    assert torch.equal(a.reset(), b.reset())
    pilot = pilots.NetPilot(a, seed=9)                                  # launch by launch, rows binned by the world's kernels
    bank = pilots.PolicyBank.random_init(b.device, seed=9, max_rows=N * 6)
    rng = np.random.default_rng(3)
    dones = 0
    for step in range(steps):
        cmd = torch.from_numpy(rng.integers(0, 3, (N, nA)).astype(np.int8)).cuda()
        want = macro_step(a, cmd, pilot)
        got = b.hl_step_nets(bank, cmd)
        torch.cuda.synchronize()
        assert b.hl_step_nets_ok(), "a grid barrier timed out"
        for x, y, name in zip(got, want, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        assert np.array_equal(a.event_masks(), b.event_masks()), f"step {step}: event masks"
        for x, y in zip(a.eval_info(), b.eval_info()):
            assert torch.equal(x, y), f"step {step}: eval counters"
        dones += int(want[3].sum())
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), f"final state {k}"
    assert a.hl_tick_count() == b.hl_tick_count()
    for x, y in zip(a.episode_stats(), b.episode_stats()):
        assert torch.equal(x, y)
    if N >= 200:
        assert dones > 0
    pilot.close()
    bank.close()
