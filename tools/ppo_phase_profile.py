"""per-phase cycle split of hh_k_policy_ppo's actor and critic tiles (needs a -DHHP_PROFILE build:
   hipcc <HIP_FLAGS of __graft_entry__.py> -DHHP_PROFILE hhmarl_2d_amd/csrc/hh_world.hip -o hhmarl_2d_amd/lib/prof_policy.so;
   run with HH_WORLD_LIB=hhmarl_2d_amd/lib/prof_policy.so)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hhmarl_2d_amd import _lib as L, pilots  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mode = sys.argv[2] if len(sys.argv) > 2 else "fight"
bank = pilots.PolicyBank.trainable_init(torch.device("cuda", 0), mode=mode, seed=3, max_rows=2 * N)
obs = torch.rand((N, 2, 30), device="cuda")
b = (pilots.SEL_FIGHT1, pilots.SEL_FIGHT2) if mode == "fight" else (pilots.SEL_ESC1, pilots.SEL_ESC2)
sel = torch.tensor(b, dtype=torch.uint8, device="cuda").repeat(N, 1).contiguous()
bank.sample(obs, sel, greedy=True)
for _ in range(10):
    bank.sample(obs, None, greedy=True)
out16, out = (C.c_ulonglong * 16)(), (C.c_ulonglong * 32)()
L.lib().hh_policy_prof_read(out16, 1)
n = 30
for _ in range(n):
    bank.sample(obs, None, greedy=True)
L.lib().hh_policy_prof_read_ppo.argtypes = [C.c_void_p]
L.lib().hh_policy_prof_read_ppo(out)
tiles = n * 2 * N / 32
names = ["rows + input gather", "L1 gemm", "L1 epilogue + barrier", "attention block", "L2 gemm", "L2 tanh + head MFMAs", "barrier (Z dead)", "partials + barrier", "logits + draw / value"]
for kind, base in (("actor", 0), ("critic", 16)):
    tot = sum(out[base:base + 9])
    print(f"-- {kind} tiles: {tot / tiles:.0f} ticks per tile (100 MHz: x 10 ns)")
    for k, nm in enumerate(names):
        print(f"   {nm:26s} {out[base + k] / tiles:8.0f}  {100.0 * out[base + k] / tot:5.1f} %")
