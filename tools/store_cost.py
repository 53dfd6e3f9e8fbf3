"""Tuning probe: time the 2-vs-2 rollout launch with and without the per-tick output stores."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 250
w = World(make_config(n_arenas=N, level=3, seed=1234, auto_reset=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
act = (torch.rand((T, N, 2, 4), device="cuda") * hi).to(torch.int8)
out = w.alloc_outputs(T)
for want in (True, False, True, False):
    for _ in range(3): w.rollout(act, out=out, want_obs=want)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): w.rollout(act, out=out, want_obs=want)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"want_obs={want}: {dt*1e3:.3f} ms / {T} ticks -> {N*T/dt/1e6:.1f} M env-steps/s")
