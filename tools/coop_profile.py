"""where the cooperative commander step (hh_hl_step_nets) spends block 0's time: world phases / policy phases / grid barriers
(needs lib/prof_coop.so built with -DHH_COOP_PROFILE)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HH_WORLD_LIB"] = os.path.join(ROOT, "hhmarl_2d_amd", "lib", "prof_coop.so")
import torch
from hhmarl_2d_amd import _lib, pilots
from hhmarl_2d_amd.world import World, make_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = World(make_config(n_arenas=N, env_kind=1, seed=1234, auto_reset=True)); w.reset()
bank = pilots.PolicyBank.random_init(w.device, seed=1, max_rows=N * 6)
cmd = (torch.rand((N, 3), device="cuda") * 3).to(torch.int8).contiguous()
out = w.alloc_outputs()
for _ in range(3): w.hl_step_nets(bank, cmd, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); w.hl_step_nets(bank, cmd, out); e1.record(); torch.cuda.synchronize()
buf = (C.c_int32 * 3)()
L = _lib.lib(); L.hh_coop_prof_read.argtypes = [C.c_void_p, C.c_void_p]; L.hh_coop_prof_read(w.h, buf)
tot = sum(buf)
print(f"N={N}: step {e0.elapsed_time(e1)*1e3:.0f} us; block 0 (100 MHz ticks -> us): world {buf[0]/100:.0f}  policy {buf[1]/100:.0f}  barriers {buf[2]/100:.0f}  total {tot/100:.0f}; ok={w.hl_step_nets_ok()}")
