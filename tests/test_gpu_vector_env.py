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
