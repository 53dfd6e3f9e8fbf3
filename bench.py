#!/usr/bin/env python
"""bench.py — env-steps/s of the MI355X air-combat world on BASELINE.json's configs[1]:
4096 arenas x 2-vs-2 fight level 3 (scripted opponent), random actions, auto-reset.

A "step" is one pass of the hot path over one batch: every arena takes one LowLevelEnv.step()
(action decode, scripted opponents, do_tick, rewards, done, observation, auto-reset), with the
action tape and all outputs resident in HBM.  Steps are issued in chunks of --chunk ticks per
persistent-kernel launch (hh_rollout); every tick writes its obs/reward/done rows.

    python bench.py                     # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, the
algorithmic bytes of SURVEY.md §8d over the measured kernel time) and `cpu_baseline` (the CPU
oracle = a C port of the reference path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_2V2_STEP = 1113   # state r/w 2*448 + actions 8 + obs 200 + reward 8 + done 1
ALGO_BYTES_3V3_CMD_STEP = 27600  # 13.2 ticks x 2056 B + commander obs 408 + actions/rewards/done
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(n_arenas, level, seed, budget_s=12.0):
    """The oracle (plain-C port of the reference path, OpenMP over arenas) on the host cores.
    Test infrastructure used only as a reported baseline, never as the measured product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0))
    w = O.OracleWorld(O.make_config(n_arenas=n_arenas, level=level, seed=seed, auto_reset=True))
    w.reset()
    rng = np.random.default_rng(seed)
    T = 25
    act = np.zeros((T, n_arenas, w.n_ctrl, 4), dtype=np.int8)
    act[..., 0] = rng.integers(0, 13, act.shape[:-1]); act[..., 1] = rng.integers(0, 9, act.shape[:-1])
    act[..., 2] = rng.integers(0, 2, act.shape[:-1]); act[..., 3] = rng.integers(0, 2, act.shape[:-1])
    w.rollout(act[:2])  # warm
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        w.rollout(act)
        steps += T
    dt = time.perf_counter() - t0
    return {"value": n_arenas * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_arenas} arenas x {steps} ticks, same config/seed/action distribution, OpenMP over arenas"}


def init_dist(rank, local_rank, world):
    """one process per GPU over RCCL (backend "nccl" is RCCL on ROCm) whenever launched by torch.distributed.run"""
    if "RANK" not in os.environ:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    return dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--arenas", type=int, default=4096, help="arenas per GPU (configs[1]: 4096)")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=250, help="ticks per persistent-kernel launch")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["low", "rollout", "hier"], default="low",
                    help="low: BASELINE configs[1] (default, the headline).  rollout: configs[2], every tick a random-init fight policy "
                         "(batched torch MLP on the same GPU) maps the observations to the next actions (use --arenas 16384).  hier: "
                         "configs[3], 3-vs-3 HighLevelEnv commander steps (use --arenas 8192); a step is one commander step = 16 "
                         "sub-steps with pilot actions")
    ap.add_argument("--pilot", choices=["tape", "random", "mlp"], default="tape",
                    help="hier: uniform actions from a pre-resident tape (default), drawn by torch kernels inside the step, or random-init MLP pilots")
    ap.add_argument("--no-graph", action="store_true", help="hier: launch the macro step eagerly instead of replaying a HIP graph")
    args = ap.parse_args()
    if args.workload == "hier":
        return main_hier(args)
    if args.workload == "rollout":
        return main_policy_rollout(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = init_dist(rank, local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas
    sw = ShardedWorld(dict(n_arenas=N, level=args.level, seed=args.seed, auto_reset=True), rank=rank, world_size=world,
                      device=local_rank)
    w = sw.world
    w.reset()
    chunk = max(1, min(args.chunk, args.steps))
    # action tape resident in HBM before the timed region: i.i.d. uniform MultiDiscrete([13,9,2,2])
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + rank)
    n_tape = 4  # distinct chunks of actions, cycled
    hi = torch.tensor([13, 9, 2, 2], device=dev)
    tape = (torch.rand((n_tape, chunk, N, w.n_ctrl, 4), device=dev, generator=gen) * hi).to(torch.int8).contiguous()
    out = w.alloc_outputs(chunk)

    def run(n_steps, timed):
        done_steps, launches, k = 0, 0, 0
        evs = []
        while done_steps < n_steps:
            T = min(chunk, n_steps - done_steps)
            if timed:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            w.rollout(tape[k % n_tape][:T], out=tuple(o[:T] for o in out))
            if timed:
                e1.record()
                evs.append((e0, e1, T))
            sw.log_episode_stats()   # RCCL all-gather of episode returns (logging only)
            done_steps += T
            launches += 1
            k += 1
        return evs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    evs = run(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: average launch duration from HIP events on the launch stream
    full = [(a.elapsed_time(b) * 1e-3, T) for a, b, T in evs if T == chunk] or [(a.elapsed_time(b) * 1e-3, T) for a, b, T in evs]
    avg_launch_s = sum(x for x, _ in full) / len(full)
    T_launch = full[0][1]
    bytes_per_launch = ALGO_BYTES_2V2_STEP * N * T_launch
    achieved = bytes_per_launch / avg_launch_s / 1e9

    # HBM bytes per launch from the PMC counters of a separate rocprofv3 run of this same command
    # (tools/prof_pmc.sh -> profiles/*_traffic.json; counters cannot be read from inside the process)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
        if tj.get("arenas") == N and tj.get("ticks_per_launch") == T_launch:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    value = N * world * args.steps / dt
    two = os.environ.get("HH_FORCE_W") == "2" or (os.environ.get("HH_FORCE_W") in (None, "", "0") and (N + 15) // 16 > 1024)
    pair = not two and os.environ.get("HH_NO_TWO") != "1" and (N + 15) // 16 <= 512
    kname = ("hh_k_world<4,64,%d,false>" % (2 if two else 1)) if os.environ.get("HH_NO_QUAD") == "1" else \
        "hh_k_world_quad<W=%d,%s>" % (2 if two else 1, "simulation wave + output wave" if pair else "single wave")
    line = {
        "metric": "env-steps/sec (2v2)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "agent_steps_per_s": value * 2,
        "config": {"workload": f"{N} arenas/GPU x 2-vs-2 fight L{args.level} (scripted opponent), random actions, auto-reset "
                               f"(BASELINE configs[1])", "arenas_per_gpu": N, "ticks_per_launch": chunk,
                   "parallelism": f"arena-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": kname, "avg_launch_ms": avg_launch_s * 1e3,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "note": "FP64-VALU issue bound, not HBM bound: see DESIGN.md section 4 (SQ_INSTS_VALU per wave-tick x 4 cycles)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(N, args.level, args.seed)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main_policy_rollout(args):
    """BASELINE configs[2]: N arenas x 2-vs-2 fight L3 driven by a policy in the loop — per tick: observations [N, 2, 26]
    -> randomly initialised fight network (26 -> 200 -> 200 -> 13+9+2+2 logits, greedy decode as env_base.py:373-382 does)
    -> hh_step.  The policy is the caller's side of the boundary (PyTorch / rocBLAS); policy + step are captured once into a
    HIP graph and replayed per tick."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = init_dist(rank, local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas
    sw = ShardedWorld(dict(n_arenas=N, level=args.level, seed=args.seed, auto_reset=True), rank=rank, world_size=world, device=local_rank)
    w = sw.world
    obs = w.reset()
    g = torch.Generator().manual_seed(args.seed)
    net = torch.nn.Sequential(torch.nn.Linear(26, 200), torch.nn.Tanh(), torch.nn.Linear(200, 200), torch.nn.Tanh(), torch.nn.Linear(200, 26))
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.3)
    net = net.to(dev)
    out = w.alloc_outputs()
    out[0].copy_(obs)
    act = torch.zeros((N, 2, 4), dtype=torch.int8, device=dev)

    @torch.no_grad()
    def tick():
        logits = net(out[0].reshape(N * 2, 26))
        parts = logits.split((13, 9, 2, 2), dim=1)
        act.copy_(torch.stack([p_.argmax(dim=1) for p_ in parts], dim=1).to(torch.int8).reshape(N, 2, 4))
        w.step(act, out=out)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            tick()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tick()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        graph.replay()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        graph.replay()
        if k % 256 == 255:
            sw.log_episode_stats()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = N * world * args.steps / dt
    achieved = ALGO_BYTES_2V2_STEP * N * args.steps / dt / 1e9
    line = {
        "metric": "env-steps/sec (2v2, policy in the loop)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "agent_steps_per_s": value * 2,
        "config": {"workload": f"{N} arenas/GPU x 2-vs-2 fight L{args.level}, actions from a random-init fight policy (fp32 MLP 26-200-200-26 "
                               f"in PyTorch, greedy decode) evaluated every tick on the same GPU, auto-reset (BASELINE configs[2])",
                   "arenas_per_gpu": N, "ticks_per_launch": 1, "parallelism": f"arena-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "hh_k_world_quad (T = 1 per launch) + the policy's rocBLAS / elementwise kernels, one HIP graph per tick"},
    }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


PILOT_DESC = {"tape": "uniform action tape resident in HBM", "random": "uniform actions drawn by torch kernels inside the step",
              "mlp": "random-init MLP fight/escape nets"}


def main_hier(args):
    """BASELINE configs[3]/[4]: N arenas x 3-vs-3 HighLevelEnv (map 0.5, horizon 500, N_OPP_HL=2), commander
    actions uniform {0,1,2}; pilots = uniform actions (default) or random-init MLPs evaluated on the same GPU."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = init_dist(rank, local_rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.pilots import MLPPilot, RandomPilot, TapePilot
    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas
    sw = ShardedWorld(dict(n_arenas=N, env_kind=1, seed=args.seed, auto_reset=True), rank=rank, world_size=world, device=local_rank)
    w = sw.world
    w.reset()
    if args.pilot == "tape":
        pilot = TapePilot(dev, N, 6, seed=args.seed + rank)
    else:
        pilot = RandomPilot(dev, args.seed + rank) if args.pilot == "random" else MLPPilot(dev, seed=args.seed)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 17 + rank)
    steps, warm = args.steps, args.warmup
    cmds = (torch.rand((64, N, 3), device=dev, generator=gen) * 3).to(torch.int8).contiguous()
    out, pbuf = w.alloc_outputs(), w.alloc_pilot()

    # the macro step is ~34 world launches + the pilots' torch kernels: capture it once into a HIP graph
    # (launch-bound inner loop; no host synchronisation inside) and replay it per commander step
    cmd_static = cmds[0].clone()
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                macro_step(w, cmd_static, pilot, out=out, pilot_buf=pbuf)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            macro_step(w, cmd_static, pilot, out=out, pilot_buf=pbuf)

    def run(n):
        for k in range(n):
            if args.pilot == "tape":
                pilot.load(k)
            if graph is not None:
                cmd_static.copy_(cmds[k % 64])
                graph.replay()
            else:
                macro_step(w, cmds[k % 64], pilot, out=out, pilot_buf=pbuf)
            if k % 16 == 15:
                sw.log_episode_stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(warm)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run(steps)
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gpu_s = e0.elapsed_time(e1) * 1e-3
    value = N * world * steps / dt
    achieved = ALGO_BYTES_3V3_CMD_STEP * N * steps / gpu_s / 1e9
    line = {
        "metric": "commander-steps/sec (3v3 HighLevelEnv)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": steps,
        "warmup": warm, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "agent_steps_per_s": value * 3, "sim_ticks_per_s": value * 16,
        "config": {"workload": f"{N} arenas/GPU x 3-vs-3 HighLevelEnv commander steps (16 sub-steps each), uniform commander actions, "
                               f"pilots = {PILOT_DESC[args.pilot]}, "
                               f"auto-reset (BASELINE configs[3])", "arenas_per_gpu": N,
                   "parallelism": f"arena-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "hh_k_hier<6,64> (all 34 phase launches of the macro step, plus the pilots' kernels if any)",
                     "algorithmic_bytes_per_step": ALGO_BYTES_3V3_CMD_STEP * N},
    }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
