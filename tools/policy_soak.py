"""Every form of the policy kernels on the rows REAL worlds produce (not uniform noise: tests/test_policy_nets.py's world-observation test exists because
noise hid a 2e-5 defect), at scale: agents' observations of 2-vs-2 level-3 worlds in fight and in escape mode collected over many ticks, then
  * hh_policy_act in every form (tile 32 / 64, hh_k_policy_w16<4>, <8>): logits against the float64 PyTorch forward (the same
    statements as the reference's forward(), policy_nets.torch_forward) with the fp32 PyTorch forward's own distance from it beside them; greedy actions against
    the float64 arg-max wherever its top two logits are more than 2e-5 apart;
  * hh_policy_sample in both forms (hh_k_policy_ppo, hh_k_policy_w16_ppo): value and logp against float64, the drawn action against the float64 inverse CDF
    wherever the uniform is more than 1e-5 from a boundary.
usage: python tools/policy_soak.py [arenas] [ticks]      (prints one line per network and form; exit code 1 if anything exceeds 1e-5)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from hhmarl_2d_amd import pilots, policy_nets as PN  # noqa: E402
import policy_ref as PR  # noqa: E402  (oracle/policy_ref.py)
from hhmarl_2d_amd.world import World, make_config  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
TOL, SEED = 1e-5, 3
dev = torch.device("cuda", 0)
FORMS = {"tile 32": {"HH_POLICY_W": "0", "HH_POLICY_TILE": "32"}, "tile 64": {"HH_POLICY_W": "0", "HH_POLICY_TILE": "64"},
         "hh_k_policy_w16<4>": {"HH_POLICY_W": "2"}, "hh_k_policy_w16<8>": {"HH_POLICY_W": "3"}}
SAMPLERS = {"hh_k_policy_ppo": {"HH_POLICY_W": "0"}, "hh_k_policy_w16_ppo": {"HH_POLICY_W": "2"}}
bad = 0


def setenv(d):
    for k in ("HH_POLICY_W", "HH_POLICY_TILE"):
        os.environ.pop(k, None)
    os.environ.update(d)


def collect(mode):
    """[T, N, 2, 30] agent observations of a level-3 world under uniform random actions, auto-reset"""
    w = World(make_config(n_arenas=N, level=3, agent_mode=1 if mode == "escape" else 0, seed=11, auto_reset=True), device=0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    hi = torch.tensor([13, 9, 2, 2], device=dev)
    act = (torch.rand((T, N, w.n_ctrl, 4), device=dev, generator=g) * hi).to(torch.int8).contiguous()
    w.reset()
    obs = w.rollout(act)[0]
    out = torch.zeros((T, N, 2, 30), device=dev)
    out[..., : obs.shape[-1]] = obs
    w.close()
    return out


for mode, kinds, sels in (("fight", (PN.FIGHT1, PN.FIGHT2), (pilots.SEL_FIGHT1, pilots.SEL_FIGHT2)), ("escape", (PN.ESC1, PN.ESC2), (pilots.SEL_ESC1, pilots.SEL_ESC2))):
    obs = collect(mode).reshape(T * N, 2, 30).contiguous()
    R = obs.shape[0]
    live = obs.abs().sum(-1) > 0                                   # rows of dead agents are zeros: kept (the kernels see them too)
    sel = torch.tensor(sels, dtype=torch.uint8, device=dev).repeat(R, 1).contiguous()
    ref64, ref32 = [], []
    for slot, kind in enumerate(kinds):
        sd = PN.random_weights(kind, SEED)
        x = obs[:, slot].cpu()
        r64 = PR.torch_forward(kind, sd, x, dtype=torch.float64)
        ref64.append(r64)
        ref32.append(float((PR.torch_forward(kind, sd, x).double() - r64).abs().max()))
    print(f"== {mode}: {R} arena-ticks x 2 agents ({int(live.sum())} rows of live agents); fp32 PyTorch forward vs float64: " + ", ".join(f"{PN.KIND_NAMES[k]} {e:.2e}" for k, e in zip(kinds, ref32)))
    for name, env in FORMS.items():
        setenv(env)
        bank = pilots.PolicyBank.random_init(dev, seed=SEED, max_rows=2 * R)
        lg = torch.zeros((R, 2, 32), device=dev)
        act = bank.act(obs.reshape(2 * R, 30), sel.reshape(-1), logits=lg.reshape(2 * R, 32)).reshape(R, 2, 4).cpu()
        for slot, kind in enumerate(kinds):
            n_out = PN.N_OUT[kind]
            err = (lg[:, slot, :n_out].cpu().double() - ref64[slot]).abs()
            want = PR.decode(ref64[slot], n_out)
            top2 = torch.stack([p.topk(2, dim=1).values for p in ref64[slot][:, :n_out].split(PN.ACTION_SPLIT[: 4 if n_out == 26 else 3], dim=1)], dim=0)
            clear = ((top2[..., 0] - top2[..., 1]) > 2 * TOL).all(dim=0)
            agree = bool((act[:, slot][clear] == want[clear]).all())
            ok = float(err.max()) <= TOL and agree
            bad += 0 if ok else 1
            print(f"  {name:20s} {PN.KIND_NAMES[kind]:7s} logits max |err| {float(err.max()):.2e} mean {float(err.mean()):.2e}; greedy action = float64 arg-max on {int(clear.sum())} clear rows"
                  f" ({R - int(clear.sum())} near-ties skipped): {'yes' if agree else 'NO'}{'' if ok else '   <-- FAIL'}")
        bank.close()
    # the sampler: uniforms drawn here, the critic's action inputs zero (as while sampling)
    u = torch.rand((R, 2, 4), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    for name, env in SAMPLERS.items():
        setenv(env)
        bank = pilots.PolicyBank.trainable_init(dev, mode=mode, seed=SEED, max_rows=2 * R, tie_shared=False)   # one shared layer per slot: the references below are built from random_weights(kind) as they are
        lg = torch.zeros((R, 2, 32), device=dev)
        act, logp, vf = bank.sample(obs, sel, uniforms=u, logits=lg)
        torch.cuda.synchronize()
        for slot, kind in enumerate(kinds):
            n_out = PN.N_OUT[kind]
            sd, csd = PN.random_weights(kind, SEED), PN.random_critic_weights(kind, SEED)
            x, x2 = obs[:, slot].cpu(), obs[:, 1 - slot].cpu()
            z = torch.zeros((R, 4))
            v64 = PR.torch_value(kind, sd, csd, x, z, x2, z, dtype=torch.float64)
            a64, lp64, margin = PR.inverse_cdf_actions(ref64[slot].numpy(), u[:, slot].cpu().numpy(), n_out)
            clear = margin > TOL
            e_v = float((vf[:, slot].cpu().double() - v64).abs().max())
            e_lp = float(np.abs(logp[:, slot].cpu().numpy().astype(np.float64) - lp64)[clear].max())
            agree = bool(np.array_equal(act[:, slot].cpu().numpy()[clear], a64[clear]))
            ok = e_v <= TOL and e_lp <= TOL and agree
            bad += 0 if ok else 1
            print(f"  {name:20s} {PN.KIND_NAMES[kind]:7s} value max |err| {e_v:.2e}; logp max |err| {e_lp:.2e}; drawn action = float64 inverse CDF on {int(clear.sum())} clear rows"
                  f" ({R - int(clear.sum())} skipped): {'yes' if agree else 'NO'}{'' if ok else '   <-- FAIL'}")
        bank.close()
print("policy soak:", "all forms within 1e-5 of the float64 forward" if not bad else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
