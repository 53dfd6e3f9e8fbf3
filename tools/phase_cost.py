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

if os.environ.get("HH_WORLD_LIB", "").endswith("prof_phases.so"):   # -DHH_PROFILE_PHASES build: the markers of hh_k_hier
    buf = (C.c_ulonglong * 16)()
    lib.hh_prof_read(buf, 1)
    g.replay(); torch.cuda.synchronize()
    lib.hh_prof_read(buf, 0)
    launches = buf[6] + buf[7]
    names = ["loads requested", "state arrived + published", "pair table (not in HL_TICK)", "phase body", "pilot rows staged + stored", "state stores issued"]
    for k, nm in enumerate(names):
        print(f"  {nm:32s} {buf[k] / launches:9.0f} cycles per wave-launch")
    print(f"  {'sum':32s} {sum(buf[:6]) / launches:9.0f} cycles per wave-launch ({launches} wave-launches)")
