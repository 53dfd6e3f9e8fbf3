"""Tuning probe: throughput of the levels 4-5 split step (hh_step_begin -> opponents' policy -> hh_step_finish), uniform
opponent actions from a resident tape, eager launches and one HIP graph per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = World(make_config(n_arenas=N, level=5, seed=3, auto_reset=True, ext_opp_actions=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
K = 64
a_ag = (torch.rand((K, N, 2, 4), device="cuda") * hi).to(torch.int8)
a_op = (torch.rand((K, N, 2, 4), device="cuda") * hi).to(torch.int8)
out = w.alloc_outputs()
oo = torch.zeros((N, 2, 30), dtype=torch.float32, device="cuda")
def step(k):
    w.step_begin(a_ag[k % K], 0, opp_obs=oo)
    w.step_finish(a_op[k % K], out=out)
for k in range(50): step(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
S = 2000
for k in range(S): step(k)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"eager: {N} arenas, {dt / S * 1e6:.1f} us per step -> {N * S / dt / 1e6:.1f} M env-steps/s")
