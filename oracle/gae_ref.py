"""
TEST INFRASTRUCTURE — what RLlib 2.4 does to the reference's step stream before PPO sees it.

ray is not installable in this image (no network) and is not part of /root/reference, so this is a restatement of the
published source of ray==2.4.0 (the version the reference pins, README.md:22) — PARITY UNPINNED against RLlib itself; the
pins are the hand-computed vectors in tests/test_rollout_post.py and the committed tests/golden/gae_vectors.npz, which
tests/ compare the HIP kernels (hh_gae_rllib / hh_gae) with.

What is restated, and where it comes from:
  * ray/rllib/evaluation/postprocessing.py `discount_cumsum(x, gamma)`:
        scipy.signal.lfilter([1], [1, float(-gamma)], x[::-1], axis=0)[::-1]
  * same file, `compute_advantages(rollout, last_r, gamma, lambda_, use_gae=True, use_critic=True)`:
        vpred_t = np.concatenate([rollout[VF_PREDS], np.array([last_r])])
        delta_t = rollout[REWARDS] + gamma * vpred_t[1:] - vpred_t[:-1]
        rollout[ADVANTAGES] = discount_cumsum(delta_t, gamma * lambda_)
        rollout[VALUE_TARGETS] = (rollout[ADVANTAGES] + rollout[VF_PREDS]).astype(np.float32)
        rollout[ADVANTAGES] = rollout[ADVANTAGES].astype(np.float32)
  * same file, `compute_gae_for_sample_batch`: last_r = 0.0 when the trajectory's last TERMINATEDS is set, else the value
    function at the last observation.
  * ray/rllib/evaluation/sampler.py `_process_observations` (the enable_connectors=False path the reference selects,
    train_hetero.py:212): one row per agent id present in the OBSERVATION dict, its reward `rewards[env_id].get(agent_id, 0.0)`,
    agent_terminated = terminateds["__all__"] or terminateds.get(agent_id).
  * the reference's side of it: observations for every agent id on every step, zeros for dead ones (envs/env_hetero.py:65-103,
    env_hier.py:49-98); rewards only for ids alive at step start (env_hetero.py:217-223; every id in HighLevelEnv,
    env_hier.py:154,188); terminateds is truncateds == {"__all__": done} (env_base.py:108).  Hence: every agent's trajectory spans
    the whole episode, dead agents' rows carry reward 0.0, nothing is masked, and last_r = 0.0 at every episode end (also at the
    horizon).  train_hetero.py:216: gamma 0.99, lambda_ 0.95; train_hier.py:186: gamma 0.99, lambda_ = RLlib's default 1.0.
"""
import numpy as np
import scipy.signal


def discount_cumsum(x, gamma):
    return scipy.signal.lfilter([1], [1, float(-gamma)], x[::-1], axis=0)[::-1]


def compute_advantages(rewards, vf_preds, last_r, gamma=0.9, lambda_=1.0):
    """one agent's trajectory of one episode (float32 arrays) -> (advantages float32, value_targets float32)"""
    vpred_t = np.concatenate([vf_preds, np.array([last_r])])
    delta_t = rewards + gamma * vpred_t[1:] - vpred_t[:-1]
    adv = discount_cumsum(delta_t, gamma * lambda_)
    value_targets = (adv + vf_preds).astype(np.float32)
    return adv.astype(np.float32), value_targets


def rllib_stream(reward, valid, value, done, gamma, lambda_):
    """the [T, N, nA] tensors of a rollout with auto-reset, cut into per-agent per-episode trajectories the way RLlib's sampler
    collects them: missing reward key -> 0.0, every row kept, last_r = 0.0 at a done row; the trailing fragment of an arena
    (episode still running at the end of the window) bootstraps from value[T] (RLlib's rule for a truncated trajectory)."""
    T, N, nA = reward.shape
    r = np.where(valid > 0, reward, np.float32(0.0)).astype(np.float32)
    adv = np.zeros((T, N, nA), dtype=np.float32)
    ret = np.zeros((T, N, nA), dtype=np.float32)
    for n in range(N):
        ends = list(np.nonzero(done[:, n])[0])
        start = 0
        for e in ends + ([T - 1] if (not ends or ends[-1] != T - 1) else []):
            complete = bool(done[e, n])
            for a in range(nA):
                last_r = 0.0 if complete else float(value[T, n, a])
                adv[start:e + 1, n, a], ret[start:e + 1, n, a] = compute_advantages(r[start:e + 1, n, a], value[start:e + 1, n, a], last_r, gamma, lambda_)
            start = e + 1
    return adv, ret


def masked_stream(reward, valid, value, done, gamma, lambda_):
    """hh_gae's own (pre-round-3) convention, kept for callers that drop dead agents' rows: a row without a reward key has
    advantage 0 and return 0 and does not propagate; float32 throughout"""
    T = reward.shape[0]
    adv = np.zeros_like(reward)
    ret = np.zeros_like(reward)
    a_next = np.zeros(reward.shape[1:], dtype=np.float32)
    v_next = value[T]
    for t in range(T - 1, -1, -1):
        nd = (1.0 - done[t].astype(np.float32))[:, None]
        delta = reward[t] + np.float32(gamma) * v_next * nd - value[t]
        a = delta + np.float32(gamma) * np.float32(lambda_) * nd * a_next
        a = np.where(valid[t] > 0, a, 0).astype(np.float32)
        adv[t] = a
        ret[t] = np.where(valid[t] > 0, a + value[t], 0)
        a_next = a
        v_next = value[t]
    return adv, ret
