"""Tuning probe: what a HighLevelEnv phase launch costs with and without its pilot-observation output (NULL pilot_obs), against the
per-sub-step cost of the one-launch macro step.  usage: phase_cost.py [arenas]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hhmarl_2d_amd import _lib as L
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = World(make_config(n_arenas=N, env_kind=1, seed=3, auto_reset=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
tape = (torch.rand((16, N, 6, 4), device="cuda") * hi).to(torch.int8)
cmd = torch.ones((N, 3), dtype=torch.int8, device="cuda")
po, pm = w.alloc_pilot()
out = w.alloc_outputs()
lib = L.lib()
p = lambda t: C.c_void_p(t.data_ptr())


def step(with_obs):
    a, b = (p(po), p(pm)) if with_obs else (None, None)
    st = w._stream()   # the stream current NOW (the capturing one)
    lib.hh_hl_begin(w.h, p(cmd), a, b, st)
    for k in range(16):
        lib.hh_hl_agents_act(w.h, p(tape[k]), a, b, st)
        lib.hh_hl_tick(w.h, p(tape[k]), a, b, None, st)
    lib.hh_hl_end(w.h, p(out[0]), p(out[1]), p(out[2]), p(out[3]), st)


for with_obs in (True, False):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(with_obs); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step(with_obs)
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"phase path, pilot observations {'written' if with_obs else 'not requested'}: {dt * 1e6:.0f} us per commander step = {dt * 1e6 / 34:.1f} us per launch")
