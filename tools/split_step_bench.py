"""Tuning probe: throughput of the levels 4-5 split step (hh_step_begin -> opponents' policy -> hh_step_finish): opponents' actions
from a resident tape (the world kernels alone) and from the frozen Fight / Esc networks (pilots.OpponentNets), eager launches and one
HIP graph per step.  usage: split_step_bench.py [arenas] [level]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hhmarl_2d_amd import pilots
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = World(make_config(n_arenas=N, level=level, seed=3, auto_reset=True, ext_opp_actions=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
K = 64
a_ag = (torch.rand((K, N, 2, 4), device="cuda") * hi).to(torch.int8)
a_op = (torch.rand((K, N, 2, 4), device="cuda") * hi).to(torch.int8)
out = w.alloc_outputs()
oo = torch.zeros((N, 2, 30), dtype=torch.float32, device="cuda")
nets = pilots.OpponentNets(w, seed=1)
mode = -1 if level == 5 else 0     # level 5: every arena observes in the mode of its own per-episode draw


def step_tape(k):
    w.step_begin(a_ag[k % K], mode, opp_obs=oo)
    w.step_finish(a_op[k % K], out=out)


def step_nets(k):
    w.step_begin(a_ag[k % K], mode, opp_obs=oo)
    w.step_finish(nets(oo), out=out)


def timed(fn, S):
    for k in range(20): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(S): fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / S


for name, fn in (("tape", step_tape), ("nets", step_nets)):
    dt = timed(fn, 1000)
    print(f"{name} eager: {N} arenas level {level}, {dt * 1e6:.1f} us per step -> {N / dt / 1e6:.1f} M env-steps/s")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(3): fn(k)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for k in range(K): fn(k)
    dt = timed(lambda k: g.replay(), 30) / K
    print(f"{name} graph: {N} arenas level {level}, {dt * 1e6:.1f} us per step -> {N / dt / 1e6:.1f} M env-steps/s")
