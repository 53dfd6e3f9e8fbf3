"""hh_policy_sample alone and inside the device-resident PPO rollout (BASELINE configs[2]): microseconds per call with / without the
value branch, greedy hh_policy_act beside it, and env-steps/s of `PPORollout.collect` (2 T + 2 launches in one HIP graph).
    python tools/ppo_bench.py [arenas] [T] [level: 3 | 4 | 5 — at 4 and 5 the frozen opponents' networks run between the two halves of every step]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hhmarl_2d_amd import pilots  # noqa: E402
from hhmarl_2d_amd.rollout import PPORollout  # noqa: E402
from hhmarl_2d_amd.world import World, make_config  # noqa: E402


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    w = World(make_config(n_arenas=N, level=level, seed=1, auto_reset=True, ext_opp_actions=level >= 4), device=0)
    bank = pilots.PolicyBank.trainable_init(w.device, seed=0, max_rows=2 * N)
    obs = w.reset()
    sel = torch.tensor([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2], dtype=torch.uint8, device=w.device).repeat(N, 1).contiguous()
    bank.sample(obs, sel, world=w)
    act = torch.zeros((N, 2, 4), dtype=torch.int8, device=w.device)
    lp, vf = torch.zeros((N, 2), device=w.device), torch.zeros((N, 2), device=w.device)
    us_full = timed(lambda: bank.sample(obs, None, world=w, actions=act, logp=lp, vf=vf))
    us_actor = timed(lambda: bank.sample(obs, None, world=w, actions=act, logp=lp, want_vf=False))
    bank.act(obs, sel, act)
    us_greedy = timed(lambda: bank.act(obs, None, act))
    print(f"{2 * N} rows: hh_policy_sample actor + value {us_full:.1f} us | actor + draw only {us_actor:.1f} us | hh_policy_act (greedy, frozen form) {us_greedy:.1f} us")
    ro = PPORollout(w, bank, T, opponents=pilots.OpponentNets(w, seed=2, skip_first=False) if level >= 4 else None)
    ro.collect(); ro.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ro.collect()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"PPORollout.collect (level {level}): {N} arenas x {T} ticks in {dt * 1e3:.2f} ms = {N * T / dt:.3e} env-steps/s ({dt / T * 1e6:.1f} us per tick)")


if __name__ == "__main__":
    main()
