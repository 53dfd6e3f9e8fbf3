import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
os.environ["HH_POLICY_W"] = sys.argv[1] if len(sys.argv) > 1 else "2"
from hhmarl_2d_amd.world import World, make_config
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from hhmarl_2d_amd import pilots, policy_nets as PN
import policy_ref as PR  # noqa: E402  (oracle/policy_ref.py)
N = 16384
MODE = sys.argv[3] if len(sys.argv) > 3 else "fight"
from hhmarl_2d_amd import _lib as L
w = World(make_config(n_arenas=N, level=3, seed=77, arena_offset=1000, auto_reset=True, agent_mode=L.MODE_FIGHT if MODE == "fight" else L.MODE_ESCAPE), device=0)
bank = pilots.PolicyBank.trainable_init(torch.device("cuda", 0), mode=MODE, seed=5, max_rows=2 * N)
obs = w.reset()
rng = np.random.default_rng(1)
for _ in range(3):
    a = torch.from_numpy(np.stack([rng.integers(0, 13, (N, 2)), rng.integers(0, 9, (N, 2)), rng.integers(0, 2, (N, 2)), rng.integers(0, 2, (N, 2))], axis=-1).astype(np.int8)).cuda()
    obs = w.step(a)[0]
sel = torch.tensor([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2] if MODE == "fight" else [pilots.SEL_ESC1, pilots.SEL_ESC2], dtype=torch.uint8, device="cuda").repeat(N, 1).contiguous()
logits = torch.zeros((N, 2, 32), dtype=torch.float32, device="cuda")
if len(sys.argv) > 2 and sys.argv[2] == "greedy":
    bank.act(obs, sel, logits=logits)
else:
    act, logp, vf = bank.sample(obs, sel, world=w, logits=logits)
torch.cuda.synchronize()
o = obs.cpu()
for slot, kind in enumerate((PN.FIGHT1, PN.FIGHT2) if MODE == "fight" else (PN.ESC1, PN.ESC2)):
    sd = PN.random_weights(kind, 5)
    ref = PR.torch_forward(kind, sd, o[:, slot])
    ref64 = PR.torch_forward(kind, {k: v.astype(np.float64) for k, v in sd.items()}, o[:, slot].double()) if False else None
    err = (logits[:, slot, :PN.N_OUT[kind]].cpu() - ref).abs()
    bad = (err.max(dim=1).values > 5e-6).nonzero().flatten()
    print(PN.KIND_NAMES[kind], "max err", float(err.max()), "rows > 5e-6:", len(bad))
    for r in bad[:4]:
        print("  row", int(r), "err", float(err[r].max()), "obs", o[r, slot].numpy().round(4))
