"""Rollout post-processing next to the environment: the centralised-critic input
packing of train_hetero.py:113-181 and GAE (train_hetero.py:216) on device tensors."""
import ctypes as C

import torch

from . import _lib as L

ACTION_DIM_AC1, ACTION_DIM_AC2 = 4, 3


def _p(t):
    return C.c_void_p(t.data_ptr())


def gae(reward, value, valid, done, gamma=0.99, lam=0.95):
    """reward, valid [T,N,nA]; value [T+1,N,nA]; done [T,N] (device tensors) -> (advantages, returns)"""
    T, N, nA = reward.shape
    assert value.shape == (T + 1, N, nA) and valid.shape == reward.shape and done.shape == (T, N)
    reward, value = reward.contiguous().float(), value.contiguous().float()
    valid, done = valid.contiguous().to(torch.uint8), done.contiguous().to(torch.uint8)
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    st = C.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream)
    L.check(L.lib().hh_gae(T, N, nA, _p(reward), _p(value), _p(valid), _p(done), float(gamma), float(lam), _p(adv), _p(ret), st))
    return adv, ret


def gae_rllib(reward, value, done, gamma=0.99, lam=0.95):
    """Advantages / value targets the way RLlib 2.4 computes them for the reference's step stream (hh_gae_rllib): rows of agents
    without a reward key stay in with reward 0.0 (nothing is masked), last_r = 0.0 at every episode end, float64 discounted sum.
    train_hetero.py:216 uses lam = 0.95, train_hier.py:186 RLlib's default lam = 1.0.
    reward [T,N,nA] (0.0 where the world reported reward_valid = 0); value [T+1,N,nA]; done [T,N] -> (advantages, value targets)"""
    T, N, nA = reward.shape
    assert value.shape == (T + 1, N, nA) and done.shape == (T, N)
    reward, value = reward.contiguous().float(), value.contiguous().float()
    done = done.contiguous().to(torch.uint8)
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    st = C.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream)
    L.check(L.lib().hh_gae_rllib(T, N, nA, _p(reward), _p(value), _p(done), float(gamma), float(lam), _p(adv), _p(ret), st))
    return adv, ret


def central_critic_inputs(obs, actions):
    """The four blocks the reference's centralised critic sees per agent (central_critic_observer +
    on_postprocess_trajectory, train_hetero.py:113-181), for the 2-vs-2 low-level setting:
      agent 1: obs_1_own = obs[1] (26|30), obs_2 = obs[2] (24|29), act_1_own = own action (4), act_2 = friend's (3)
      agent 2: obs_1_own = obs[2],         obs_2 = obs[1],          act_1_own = own (3),        act_2 = friend's (4)
    with the heading / speed components scaled by 1/12 and 1/8 (train_hetero.py:143-160).
    obs [..., 2, D] (zero padded rows as the world emits them), actions int8 [..., 2, 4].
    Returns a dict per agent of float32 tensors with the reference's widths."""
    d1 = obs.shape[-1]
    d2 = d1 - 2 if d1 == 26 else d1 - 1          # 26/24 fight, 30/29 escape
    # the reference divides the integer actions in float64 and stores into the float32 batch (train_hetero.py:143-160): the same here
    # on any device (a float32 division on the GPU is not correctly rounded in every PyTorch build, the float64 one is)
    a = actions.double()
    scaled = torch.stack([a[..., 0] / 12.0, a[..., 1] / 8.0, a[..., 2], a[..., 3]], dim=-1).float()
    o1, o2 = obs[..., 0, :d1], obs[..., 1, :d2]
    a1, a2 = scaled[..., 0, :ACTION_DIM_AC1], scaled[..., 1, :ACTION_DIM_AC2]
    return {1: {"obs_1_own": o1, "obs_2": o2, "act_1_own": a1, "act_2": a2},
            2: {"obs_1_own": o2, "obs_2": o1, "act_1_own": a2, "act_2": a1}}


def central_critic_rows(obs, actions, agent):
    """The flattened CUR_OBS rows the reference's critic is trained on, for the 2-vs-2 low-level setting: RLlib flattens the
    observer's Dict in sorted key order (act_1_own, act_2, obs_1_own, obs_2) and `on_postprocess_trajectory`
    (train_hetero.py:120-160) then writes the own and the friend's actions into the first 7 columns, heading / speed components
    scaled by 1/12 and 1/8.  Agent 1 (type 1): [own act 4 | friend act 3 | own obs 26|30 | friend obs 24|29]; agent 2 (type 2):
    [own act 3 | friend act 4 | own obs 24|29 | friend obs 26|30].  obs [..., 2, D] as the world emits it (zero padded), actions
    int8 [..., 2, 4] -> float32 [..., 57] (fight) / [..., 66] (escape).  Pinned by tests/golden/critic_packing.npz (recorded
    from the reference's own callbacks)."""
    c = central_critic_inputs(obs, actions)[agent]
    return torch.cat([c["act_1_own"], c["act_2"], c["obs_1_own"], c["obs_2"]], dim=-1).float()


def central_critic_rows_hl(obs, actions, agent):
    """train_hier.py:100-165 for the 3-vs-3 commander: sorted keys (act_1_own, act_2, act_3, obs_1_own, obs_2, obs_3); the own
    action and the other two agents' actions (ascending id) divided by N_OPP_HL = 2 in columns 0..2.  obs [..., 3, 34],
    actions int8 [..., 3] -> float32 [..., 105]."""
    others = [i for i in (1, 2, 3) if i != agent]
    order = [agent] + others
    a = torch.stack([actions[..., i - 1].double() / 2.0 for i in order], dim=-1).float()
    return torch.cat([a] + [obs[..., i - 1, :].float() for i in order], dim=-1)


def episode_segments(done):
    """batch_mode="complete_episodes" (train_hetero.py:212) on the [T, N] done flags of a rollout with auto-reset:
    -> (segment id per row [T, N], starting at 0 per arena and increasing after every done row;
        complete [T, N] bool: the row belongs to an episode that also ENDS inside this rollout — the rows RLlib would train on;
        the trailing fragment of every arena is carried into the next rollout instead)"""
    d = done.to(torch.int64)
    seg = torch.cumsum(d, dim=0) - d                     # episodes finished strictly before this row
    n_done = d.sum(dim=0, keepdim=True)                  # episodes that end inside the rollout, per arena
    return seg, seg < n_done
