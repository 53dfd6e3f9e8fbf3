"""hh_policy_act alone: microseconds per call and fp32-equivalent TFLOP/s for R rows of Fight1 / Fight2 (the configs[2] mix), per tile
width of the split-fp16 kernel (HH_POLICY_TILE is read at hh_policy_create).  usage: python tools/policy_bench.py [rows] [tiles...]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hhmarl_2d_amd import pilots, policy_nets as PN  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
tiles = sys.argv[2:] or ["32", "64", "0"]
obs = torch.rand((R, 26), device="cuda")
sel = torch.tensor([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2], dtype=torch.uint8, device="cuda").repeat(R // 2).contiguous()
flops = R / 2 * (PN.flops_per_row(PN.FIGHT1) + PN.flops_per_row(PN.FIGHT2))
for t in tiles:
    os.environ["HH_POLICY_TILE"] = t
    bank = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=3, max_rows=R)
    bank.act(obs, sel)
    for _ in range(20):
        bank.act(obs, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        bank.act(obs, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"tile {t:>2s}: {us:8.2f} us per call of {R} rows = {flops / us * 1e-6:7.1f} TFLOP/s fp32-equivalent")
    bank.close()
