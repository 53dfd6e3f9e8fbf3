"""
TEST INFRASTRUCTURE — plain PyTorch fp32 / numpy float64 restatements of the policy networks' arithmetic: the numerics reference
the fused HIP kernels (hh_policy_kernel*.h) are compared with.  Only tests/, tools/ and oracle/gen_policy_golden.py import this; the
product package (hhmarl_2d_amd/) never does.  Pinned itself by tests/golden/policy_nets.npz / policy_value.npz, which hold the outputs
of the reference's OWN Fight1/Fight2/Esc1/Esc2 classes (models/ac_models_hetero.py:29-404; oracle/gen_policy_golden.py).
"""
import numpy as np

from hhmarl_2d_amd.policy_nets import ACTION_SPLIT, CRITIC_DIMS, HAS_ATT, INPUTS, N_OUT, OBS_DIM  # noqa: F401  (architecture description only)


def torch_forward(kind, sd, obs, dtype=None):
    """plain PyTorch fp32 actor forward (same op as the HIP kernel, statement order of the reference's forward()):
    obs float32 [R, >= OBS_DIM[kind]] -> logits float32 [R, N_OUT[kind]].  dtype = torch.float64: the same arithmetic in double (the
    yardstick of tools/policy_soak.py: how far the fp32 forward itself is from the real-number result)"""
    import torch
    import torch.nn.functional as F
    dtype = dtype or torch.float32
    t = {k: torch.as_tensor(v, dtype=dtype, device=obs.device) for k, v in sd.items()}
    x = obs[:, :OBS_DIM[kind]].to(dtype)
    h = []
    for n, (c0, c1, _) in zip(("inp1", "inp2", "inp3"), INPUTS[kind]):
        h.append(torch.tanh(F.linear(x[:, c0:c1], t[f"{n}._model.0.weight"], t[f"{n}._model.0.bias"])))
    if HAS_ATT[kind]:
        wv, bv = t["att_act.in_proj_weight"][200:300], t["att_act.in_proj_bias"][200:300]
        att = F.linear(F.linear(h[2], wv, bv), t["att_act.out_proj.weight"], t["att_act.out_proj.bias"])
        h[2] = F.normalize(h[2] + att)
    z = torch.cat(h, dim=1)
    s = torch.tanh(F.linear(z, t["shared_layer._model.0.weight"], t["shared_layer._model.0.bias"]))
    return F.linear(s, t["act_out._model.0.weight"], t["act_out._model.0.bias"])


def decode(logits, n_out):
    """env_base.py:373-382: greedy action per MultiDiscrete component; type-2 aircraft have no 4th component (-> 0)"""
    import torch
    parts = logits[:, :n_out].split(ACTION_SPLIT[: 4 if n_out == 26 else 3], dim=1)
    act = torch.zeros((logits.shape[0], 4), dtype=torch.int8, device=logits.device)
    for i, p in enumerate(parts):
        act[:, i] = p.argmax(dim=1).to(torch.int8)
    return act


def torch_value(kind, sd, csd, obs_own, act_own, obs_2, act_2, dtype=None):
    """plain PyTorch fp32 value_function() (statement order of the reference): sd = actor tensors (for the shared layer), csd = value branch"""
    import torch
    import torch.nn.functional as F
    dev = obs_own.device
    dtype = dtype or torch.float32
    t = {k: torch.as_tensor(v, dtype=dtype, device=dev) for k, v in {**sd, **csd}.items()}
    d1, a1, d2, a2 = CRITIC_DIMS[kind]
    v1 = torch.cat((obs_own[:, :d1], act_own[:, :a1]), dim=1).to(dtype)
    v2 = torch.cat((obs_2[:, :d2], act_2[:, :a2]), dim=1).to(dtype)
    v3 = torch.cat((v1, v2), dim=1)
    if HAS_ATT[kind]:
        y = torch.cat((torch.tanh(F.linear(v1, t["v1._model.0.weight"], t["v1._model.0.bias"])),
                       torch.tanh(F.linear(v2, t["v2._model.0.weight"], t["v2._model.0.bias"]))), dim=1)
        yf = torch.tanh(F.linear(v3, t["v3._model.0.weight"], t["v3._model.0.bias"]))
        wv, bv = t["att_val.in_proj_weight"][300:450], t["att_val.in_proj_bias"][300:450]
        att = F.linear(F.linear(yf, wv, bv), t["att_val.out_proj.weight"], t["att_val.out_proj.bias"])
        y = torch.cat((y, F.normalize(yf + att)), dim=1)
    else:
        y = torch.tanh(F.linear(v3, t["inp1_val._model.0.weight"], t["inp1_val._model.0.bias"]))
    s = torch.tanh(F.linear(y, t["shared_layer._model.0.weight"], t["shared_layer._model.0.bias"]))
    return F.linear(s, t["val_out._model.0.weight"], t["val_out._model.0.bias"]).reshape(-1)


def inverse_cdf_actions(logits, u, n_out):
    """hh_policy_sample's draw restated in float64: per component the first index whose cumulative softmax exceeds u (numpy [R, >= n_out],
    u [R, 4]) -> (actions int8 [R, 4], logp float64 [R], margin float64 [R] = distance of u from the nearest cumulative boundary)"""
    lg = np.asarray(logits, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    act = np.zeros((lg.shape[0], 4), dtype=np.int8)
    logp = np.zeros(lg.shape[0])
    margin = np.full(lg.shape[0], np.inf)
    lo = 0
    for k, w in enumerate(ACTION_SPLIT[: 4 if n_out == 26 else 3]):
        seg = lg[:, lo:lo + w]
        m = seg.max(axis=1, keepdims=True)
        e = np.exp(seg - m)
        S = e.sum(axis=1, keepdims=True)
        cdf = np.cumsum(e, axis=1) / S
        a = (cdf > u[:, k:k + 1]).argmax(axis=1)
        a = np.where((cdf > u[:, k:k + 1]).any(axis=1), a, w - 1)
        act[:, k] = a
        logp += (seg[np.arange(len(a)), a] - m[:, 0]) - np.log(S[:, 0])
        margin = np.minimum(margin, np.abs(cdf[:, :-1] - u[:, k:k + 1]).min(axis=1))
        lo += w
    return act, logp, margin


def multicategorical_logp(logits, act, n_out):
    """TorchMultiCategorical.logp (ray/rllib/models/torch/torch_action_dist.py): the sum of the components' Categorical log_prob"""
    import torch
    lg = torch.as_tensor(logits, dtype=torch.float32)
    a = torch.as_tensor(np.asarray(act), dtype=torch.int64)
    parts = lg[:, :n_out].split(ACTION_SPLIT[: 4 if n_out == 26 else 3], dim=1)
    return sum(torch.distributions.Categorical(logits=p).log_prob(a[:, i]) for i, p in enumerate(parts))
