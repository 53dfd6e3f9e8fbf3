"""Per-phase cycle breakdown of the tick (tuning tool; needs lib/prof_phases.so built with -DHH_PROFILE_PHASES)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HH_WORLD_LIB", os.path.join(ROOT, "hhmarl_2d_amd", "lib", "abl_phases.so"))   # bash tools/build_variant.sh phases -DHH_PROFILE_PHASES
import torch
from hhmarl_2d_amd import _lib
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 250
w = World(make_config(n_arenas=N, level=3, seed=1234, auto_reset=True)); w.reset()
hi = torch.tensor([13, 9, 2, 2], device="cuda")
act = (torch.rand((T, N, 2, 4), device="cuda") * hi).to(torch.int8)
out = w.alloc_outputs(T)
w.rollout(act, out=out)
buf = (C.c_ulonglong * 24)()
L = _lib.lib(); L.hh_prof_read(buf, 1)
for _ in range(4): w.rollout(act, out=out)
L.hh_prof_read(buf, 0)
names = ["A2 level-3 script (opponents)", "B kinematics+move", "Q enqueue", "I drain (estimate)", "L+C+D launch/resolve/rocket move", "E rewards", "publish", "pair tables", "finish (shaping, done)", "stats/outputs/reset", "K2 observe+store", "A1 rekey + agents' action decode"]
APW = 8 if N <= 4096 and os.environ.get("HH_APW") != "16" else 16   # the form small worlds run (hh_kernels_quad.h)
waves = (N + APW - 1) // APW
tot = sum(buf[:12])
for k, nm in enumerate(names):
    print(f"{nm:36s} {buf[k] / (waves * 4 * T):9.0f} cycles/wave-tick  {100.0 * buf[k] / tot:5.1f} %")
print(f"{'total':36s} {tot / (waves * 4 * T):9.0f} cycles/wave-tick")
wt = waves * 4 * T
print(f"queue: non-empty on {buf[12] / wt:.3f} of wave-ticks; entries per wave-tick: launch {buf[13] / wt:.3f}, cannon {buf[14] / wt:.3f}, rocket fuse {buf[15] / wt:.3f}")
if buf[16] or buf[17]:   # two-wave preset instances: the output wave builds the pair table (hh_kernels_quad.h: OWT)
    on = ["wait at barrier X", "pair table", "wait at barrier Y", "observation rows + stores"]
    for k, nm in enumerate(on):
        print(f"output wave: {nm:28s} {buf[16 + k] / (waves * 4 * T):9.0f} cycles/wave-tick")
    print(f"simulation wave: barrier X wait {buf[20] / (waves * 4 * T):.0f}, barrier Y wait + table read {buf[21] / (waves * 4 * T):.0f} cycles/wave-tick")
