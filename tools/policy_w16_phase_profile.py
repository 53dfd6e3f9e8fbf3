"""phase timers of hh_k_policy_w16 (wave 0 of every tile; -DHHP_PROFILE build: bash tools/build_variant.sh prof "-DHHP_PROFILE", then
HH_WORLD_LIB=hhmarl_2d_amd/lib/abl_prof.so python tools/policy_w16_phase_profile.py [rows])"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["HH_POLICY_W"] = os.environ.get("W", "2")   # 2: 64-row tiles (two workgroups per CU), 3: 128-row tiles (eight waves)
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
torch.cuda.synchronize()
L.lib().hh_policy_prof_read(out, 0)
tiles = n * R / (128 if os.environ["HH_POLICY_W"] == "3" else 64)
names = ["prologue: rows, observation, biases, chunk 0 wait", "L1 chunk 0 (16 tiles: MFMA + tanh/split)", "barrier (chunk 1)", "L1 chunk 1", "attention block (2 chunks) + normalisation",
         "shared layer: 32 barriers (chunk waits)", "shared layer: 256 steps (4 reads + 6 MFMAs; LDS-DMA requests)", "shared layer: tanh/split of 16 fragments", "head: 8 x (global fragments + 12 MFMAs)",
         "logits through LDS + decode"]
tot = sum(out[:10])
for k, nm in enumerate(names):
    print(f"{nm:62s} {out[k] / tiles:9.0f} cycles/tile {100.0 * out[k] / tot:5.1f} %")
rt = out[15] / tiles  # s_memrealtime: 100 MHz
print(f"total {tot / tiles:.0f} shader-clock cycles per tile in {rt * 10:.0f} ns of s_memrealtime = {tot / tiles / (rt * 10):.2f} GHz while the tile ran")
