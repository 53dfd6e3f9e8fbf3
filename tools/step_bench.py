"""Tuning probe: hh_step (one tick per launch, the RLlib-facing path) eager and from a HIP graph, against hh_rollout's persistent
per-tick cost.  usage: step_bench.py [arenas]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = World(make_config(n_arenas=N, level=3, seed=3, auto_reset=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
K = 64
acts = (torch.rand((K, N, 2, 4), device="cuda") * hi).to(torch.int8)
out = w.alloc_outputs()


def timed(fn, S):
    for k in range(20): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(S): fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / S


dt = timed(lambda k: w.step(acts[k % K], out=out), 2000)
print(f"hh_step eager: {N} arenas, {dt * 1e6:.1f} us per step -> {N / dt / 1e6:.1f} M env-steps/s   [{w.kernel_name()}]")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for k in range(3): w.step(acts[k], out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for k in range(K): w.step(acts[k], out=out)
dt = timed(lambda k: g.replay(), 30) / K
print(f"hh_step graph: {N} arenas, {dt * 1e6:.1f} us per step -> {N / dt / 1e6:.1f} M env-steps/s")
tape = acts[:, None].expand(K, 1, N, 2, 4)
outs = w.alloc_outputs(K)
dt = timed(lambda k: w.rollout(acts, out=outs), 50) / K
print(f"hh_rollout ({K} ticks per launch): {dt * 1e6:.1f} us per tick -> {N / dt / 1e6:.1f} M env-steps/s")
