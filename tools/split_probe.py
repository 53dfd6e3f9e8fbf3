"""probe: K sub-worlds x (N/K) arenas, each with its own NetPilot, macro steps on K streams inside one HIP graph"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hhmarl_2d_amd.env_hier import macro_step
from hhmarl_2d_amd.pilots import NetPilot
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
EAGER = len(sys.argv) > 2 and sys.argv[2] == "eager"
KS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
for K in KS:
    n = N // K
    worlds = [World(make_config(n_arenas=n, env_kind=1, seed=1234, auto_reset=True, arena_offset=k * n)) for k in range(K)]
    pilots = [NetPilot(w, seed=1234) for w in worlds]
    for w in worlds: w.reset()
    cmd = [(torch.rand((n, 3), device="cuda") * 3).to(torch.int8).contiguous() for _ in range(K)]
    outs = [w.alloc_outputs() for w in worlds]
    pbufs = [w.alloc_pilot() for w in worlds]
    streams = [torch.cuda.Stream() for _ in range(K)]
    def step():
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                macro_step(worlds[k], cmd[k], pilots[k], out=outs[k], pilot_buf=pbufs[k])
        for k in range(K):
            cur.wait_stream(streams[k])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    class _E:
        replay = staticmethod(step)
    g = _E
    if not EAGER:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"N={N} K={K}: {dt*1e3:.3f} ms per commander step, {N/dt:.3e} commander-steps/s", flush=True)
    for p in pilots: p.close()
    for w in worlds: w.close()
