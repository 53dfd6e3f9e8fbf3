"""GPU time of one HighLevelEnv phase launch (hh_hl_agents_act, hh_hl_tick) against the world size, next to the per-sub-step time of
the one-launch macro step: tells a throughput bound (time grows with arenas) from a per-launch latency bound (it does not)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hhmarl_2d_amd.world import World, make_config

def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

for N in (64, 1024, 8192, 32768):
    w = World(make_config(n_arenas=N, env_kind=1, seed=7, auto_reset=True))
    w.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    hi = torch.tensor([13, 9, 2, 2], device="cuda")
    act = (torch.rand((N, 6, 4), device="cuda", generator=g) * hi).to(torch.int8).contiguous()
    tape = (torch.rand((16, N, 6, 4), device="cuda", generator=g) * hi).to(torch.int8).contiguous()
    cmd = (torch.rand((N, 3), device="cuda", generator=g) * 3).to(torch.int8).contiguous()
    pb = w.alloc_pilot(); out = w.alloc_outputs()
    w.hl_begin(cmd, pb)
    # graph of 8 x (act, tick): no host launch gaps in the measurement
    def pair():
        w.hl_agents_act(act, pb); w.hl_tick(act, pb, count_running=False)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        pair()
    torch.cuda.current_stream().wait_stream(s)
    w.hl_end(out)
    def step(k):
        w.hl_begin(cmd, pb)
        for _ in range(k): pair()
        w.hl_end(out)
    with torch.cuda.stream(s):
        step(1)
    torch.cuda.current_stream().wait_stream(s)
    g12, g4 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g12):
        step(12)
    with torch.cuda.graph(g4):
        step(4)
    t_pair = (timed(g12.replay, 30) - timed(g4.replay, 30)) / 8     # sub-steps 5..12 of a macro step: most arenas still inside it
    t_macro = timed(lambda: w.hl_rollout(cmd, tape, out=out), 30)
    print(f"N={N:6d}: act+tick launches {t_pair:7.1f} us per sub-step   macro step {t_macro:7.1f} us ({t_macro/13.45:5.1f} us per sub-step)   instance {w.kernel_instance(0)} / {w.kernel_instance(1)}", flush=True)
    w.close()
