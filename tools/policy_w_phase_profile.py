"""phase timers of hh_k_policy_w (wave 0 of every tile; -DHHP_PROFILE build, HH_WORLD_LIB=hhmarl_2d_amd/lib/prof_policy.so HH_POLICY_W=1)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["HH_POLICY_W"] = "1"
from hhmarl_2d_amd import _lib as L, pilots  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
bank = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=3, max_rows=R)
obs = torch.rand((R, 26), device="cuda")
sel = torch.tensor([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2], dtype=torch.uint8, device="cuda").repeat(R // 2).contiguous()
bank.act(obs, sel)
for _ in range(10):
    bank.act(obs, None)
out = (C.c_ulonglong * 16)()
L.lib().hh_policy_prof_read(out, 1)
n = 30
for _ in range(n):
    bank.act(obs, None)
L.lib().hh_policy_prof_read(out, 0)
tiles = n * R / 128
names = ["rows + obs + L1 chunk wait", "L1 (16 tiles: MFMA + epilogue)", "barrier", "attention", "barrier", "-", "L2 half 0 (MFMAs + fillers)", "L3 of the previous pair", "barrier",
         "L2 half 1 (MFMAs + LDS-DMA)", "last pair's epilogue", "barrier", "logits + decode"]
tot = sum(out[:13])
for k, nm in enumerate(names):
    print(f"{nm:34s} {out[k] / tiles:9.0f} cycles/tile {100.0 * out[k] / tot:5.1f} %")
print(f"total {tot / tiles:.0f} cycles per tile")
