"""logit error of the policy kernels (split-fp16; HH_POLICY_W / HH_POLICY_TILE pick the form) against a float64 PyTorch forward"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from hhmarl_2d_amd import pilots, policy_nets as PN  # noqa: E402
import policy_ref as PR  # noqa: E402  (oracle/policy_ref.py)

R = 8192
rng = np.random.default_rng(0)
bank = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=3, max_rows=R)
for kind, byte in ((PN.FIGHT1, pilots.SEL_FIGHT1), (PN.FIGHT2, pilots.SEL_FIGHT2), (PN.ESC1, pilots.SEL_ESC1), (PN.ESC2, pilots.SEL_ESC2)):
    obs = torch.zeros((R, 30))
    obs[:, : PN.OBS_DIM[kind]] = torch.from_numpy(rng.random((R, PN.OBS_DIM[kind])).astype(np.float32))
    sel = torch.full((R,), byte, dtype=torch.uint8, device="cuda")
    logits = torch.zeros((R, 32), device="cuda")
    bank.act(obs.cuda(), sel, logits=logits)
    sd64 = {k: v.astype(np.float64) for k, v in PN.random_weights(kind, 3).items()}
    t = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in sd64.items()}
    import torch.nn.functional as F
    x = obs[:, : PN.OBS_DIM[kind]].double()
    h = [torch.tanh(F.linear(x[:, c0:c1], t[f"{n}._model.0.weight"], t[f"{n}._model.0.bias"])) for n, (c0, c1, _) in zip(("inp1", "inp2", "inp3"), PN.INPUTS[kind])]
    if PN.HAS_ATT[kind]:
        att = F.linear(F.linear(h[2], t["att_act.in_proj_weight"][200:300], t["att_act.in_proj_bias"][200:300]), t["att_act.out_proj.weight"], t["att_act.out_proj.bias"])
        h[2] = F.normalize(h[2] + att)
    s = torch.tanh(F.linear(torch.cat(h, 1), t["shared_layer._model.0.weight"], t["shared_layer._model.0.bias"]))
    ref64 = F.linear(s, t["act_out._model.0.weight"], t["act_out._model.0.bias"])
    ref32 = PR.torch_forward(kind, PN.random_weights(kind, 3), obs)
    got = logits[:, : PN.N_OUT[kind]].cpu().double()
    print(f"{PN.KIND_NAMES[kind]:7s} kernel vs f64: max {float((got - ref64).abs().max()):.2e} mean {float((got - ref64).abs().mean()):.2e} | "
          f"torch fp32 (CPU) vs f64: max {float((ref32.double() - ref64).abs().max()):.2e} mean {float((ref32.double() - ref64).abs().mean()):.2e}")
