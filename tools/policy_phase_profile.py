"""per-phase cycle split of hh_k_policy_h (needs hhmarl_2d_amd/lib/prof_policy.so built with -DHHP_PROFILE:
   hipcc <HIP_FLAGS of __graft_entry__.py> -DHHP_PROFILE hhmarl_2d_amd/csrc/hh_world.hip -o hhmarl_2d_amd/lib/prof_policy.so;
   run with HH_WORLD_LIB=hhmarl_2d_amd/lib/prof_policy.so)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hhmarl_2d_amd import _lib as L, pilots  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
bank = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=3, max_rows=R)
obs = torch.rand((R, 26), device="cuda")
sel = torch.tensor([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2], dtype=torch.uint8, device="cuda").repeat(R // 2).contiguous()
for _ in range(20):
    bank.act(obs, sel)
out = (C.c_ulonglong * 16)()
L.lib().hh_policy_prof_read(out, 1)
n = 50
for _ in range(n):
    bank.act(obs, sel)
L.lib().hh_policy_prof_read(out, 0)
TILE = int(os.environ.get("HH_POLICY_TILE", "32"))
names = ["rows + obs gather", "L1 gemm", "L1 epilogue + barrier", "attention block", "L2 gemm", "L2 tanh", "barrier (Z dead)", "L2 store + barrier", "L3 + logits + decode"]
print("attention detail: gemm", out[9] / (n * R / TILE), "| y + row sums", out[10] / (n * R / TILE), "| barrier", out[11] / (n * R / TILE), "(the remainder of the block: normalise + store + barrier)")
tot = sum(out[:9])
tiles = n * R / TILE
for k, nm in enumerate(names):
    print(f"{nm:24s} {out[k] / tiles:10.0f} ticks/tile  {100.0 * out[k] / tot:5.1f} %")
print("total", tot / tiles, "s_memtime ticks per tile (100 MHz constant clock: x 10 ns)")
