#!/usr/bin/env python
"""bench.py — env-steps/s of the MI355X air-combat world on BASELINE.json's configs[1]:
4096 arenas x 2-vs-2 fight level 3 (scripted opponent), random actions, auto-reset.

One *step* of `--steps` / `--warmup` is ONE pass of the hot path over one batch of synthetic input = one
`hh_rollout` launch of the persistent kernel: every arena takes `--chunk` (250) consecutive
LowLevelEnv.step() calls (action decode, scripted opponents, do_tick, rewards, done, observation,
auto-reset), action tape and all outputs resident in HBM, every tick's obs/reward/done rows written.
`value` = arenas x chunk x steps x ranks / wall time = env-steps/s (LowLevelEnv.step() calls per second).

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # spawns 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # what the driver does

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM: the algorithmic
bytes of SURVEY.md §8d over the kernel time measured with HIP events; plus the FP64-issue fraction, which
is the bound this kernel actually sits on) and `cpu_baseline` (the CPU oracle = a C port of the reference
path timed live on this box's host cores, plus the reference's own Python rate measured in the build
container and committed under profiles/).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_2V2_STEP = 1113       # state r/w 2*448 + actions 8 + obs 200 + reward 8 + done 1
ALGO_BYTES_3V3_TICK = 2056       # 656 + 656 + pilot obs 720 + pilot actions 24
ALGO_BYTES_3V3_TICK_TAPE = 1336  # the same tick when the pilots' actions come from a tape: nobody builds or reads the 720 B of pilot observations
ALGO_BYTES_3V3_STATE = 656       # one arena's state record (read once and written once per commander step by the one-launch macro step)
ALGO_BYTES_3V3_CMD_FIXED = 424   # commander obs 408 + actions 3 + rewards 12 + done 1
DEFAULT_STREAMS = {"rollout": 2, "hier_net": 4, "hier_net_variants": 4}   # sub-worlds / streams of the policy-in-the-loop workloads (--streams)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0    # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md; AMD's headline figure includes 2:1 sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3     # fp32-in MFMA runs at the vector rate
FP64_ISSUE_PEAK = 3.93e13        # lane-operations/s: 78.6 TFLOP/s vector FP64 / 2 flop per FMA (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed launches (one step = one launch of --chunk ticks over all arenas)")
    ap.add_argument("--warmup", type=int, default=8, help="untimed launches after the clock spin-up")
    ap.add_argument("--arenas", type=int, default=None, help="arenas per GPU (configs[1]: 4096; rollout: 16384; hier: 8192)")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=500, help="ticks per persistent-kernel launch (= per step).  500: the driver's --steps 20 --warmup 5 then "
                                                            "times 10 000 ticks of every arena after 2 500 warm-up ticks (SURVEY.md 8d asks >= 10 000 after 1 000)")
    ap.add_argument("--one-gpu-value", type=float, default=None, help="--gpus N > 1: the `value` of a 1-GPU line of the same command; the N-GPU line then carries "
                                                                       "scaling_efficiency = value_N / (N x value_1) (weak scaling, SURVEY.md 8e)")
    ap.add_argument("--tape", choices=["keyed", "torch"], default="keyed", help="action tapes: keyed = hh_action_tape_uniform, i.i.d. uniform over MultiDiscrete([13,9,2,2]) from the "
                                                                                  "keyed RNG, key = (seed, global arena, step, agent) (SURVEY.md 8d); torch = torch.rand with a per-rank seed")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--spinup", type=float, default=0.6, help="seconds of untimed launches before the warm-up (GPU clocks ramp from idle)")
    ap.add_argument("--log-every", type=int, default=8, help="launches between logging all-gathers of episode statistics (side stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU plumbing check of the multi-rank path (gloo, no world, no kernel): ranks, barriers, max-over-ranks, gather")
    ap.add_argument("--workload", choices=["low", "rollout", "hier", "collect"], default="low",
                    help="low: BASELINE configs[1] (default, the headline).  rollout: configs[2], every tick a fight policy with the "
                         "reference's Fight1/Fight2 architecture maps the observations to the next actions (--arenas 16384).  hier: "
                         "configs[3], 3-vs-3 HighLevelEnv commander steps (--arenas 8192); a step is one commander step.  collect: configs[2] as a whole "
                         "PPO batch — PPORollout.collect of --chunk ticks (sampler + step per tick, bootstrap value, hh_gae_rllib): a step is one collect")
    ap.add_argument("--pilot", choices=["tape", "random", "mlp", "net"], default="tape",
                    help="hier: uniform actions from a pre-resident tape (default), drawn by torch kernels inside the step, random-init "
                         "MLP stand-ins in torch, or the reference's Fight/Esc architectures in the fused HIP kernel (net)")
    ap.add_argument("--ppo", action="store_true", help="rollout: the TRAINABLE policies as RLlib's sampler evaluates them (train_hetero.py:206-243) instead of a greedy "
                                                        "actor: per tick hh_policy_sample = actor forward + Categorical draw per action component (keyed RNG) + its "
                                                        "log-probability + the centralised value branch on central_critic_observer's rows, then hh_step")
    ap.add_argument("--no-graph", action="store_true", help="hier/rollout: launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--joined", action="store_true", help="hier --pilot net: ONE HIP graph whose sub-world branches join at the end of every commander step "
                                                          "(default: one graph per sub-world, each replayed on its own stream)")
    ap.add_argument("--pilot-rows", choices=["variants", "sides"], default="variants",
                    help="hier --pilot net: 'variants' = one launch + one policy call per sub-step (each opponent's row evaluated in the variants the agents' "
                         "same-sub-step weapon flags can produce: hh_hl_begin_variants / hh_hl_act_tick); 'sides' = the two-launch, two-call path (A/B; same trajectories)")
    ap.add_argument("--phases", action="store_true", help="hier with --pilot tape: the 34-launch phase path instead of the one-launch macro step")
    ap.add_argument("--streams", type=int, default=0, help="rollout / hier --pilot net: split the arenas into this many sub-worlds (disjoint global arena ids, "
                                                                 "bit-identical to one world) stepped on as many HIP streams inside the one graph, so that one sub-world's world "
                                                                 "launches run under another's policy kernel (0 = the workload's default)")
    ap.add_argument("--no-extra", action="store_true", help="default workload at 1 GPU: skip the short runs of the other single-GPU configurations "
                                                             "(BASELINE configs[2], configs[3] tape / networks) that fill line['extra']")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ multi-rank plumbing
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one process per GPU
    (the reference's analogue is num_rollout_workers, train_hetero.py:212).  Rank 0 of the child job prints the line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


class Ranks:
    """RANK / LOCAL_RANK / WORLD_SIZE from the launcher; RCCL ("nccl") on GPUs, gloo for --dry-run"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dry = args.dry_run
        self.dist = None
        if not self.dry:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        else:
            self.dev = torch.device("cpu")
        if "RANK" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("NCCL_DEBUG", "WARN")   # keep RCCL's version banner out of the output: rank 0 prints ONE line
            if self.dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x):
        if self.dist is None:
            return [x]
        t = self.torch.tensor([x], device=self.dev, dtype=self.torch.float64)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class DryWorld:
    """--dry-run stand-in: no arenas, no kernel — only so that the rank plumbing above can be exercised on a box without
    GPUs (tests/test_bench_launcher.py).  Its line is marked "dry_run": true and measures nothing."""

    def __init__(self, kw):
        import torch
        self.N, self.n_ctrl, self.n_agents, self.D, self.device = kw["n_arenas"], 2, 2, 26, torch.device("cpu")
        self.offset = kw["arena_offset"]

    def reset(self):
        return None

    def alloc_outputs(self, T=None):
        return (None, None, None, None)

    def rollout(self, act, out=None):
        time.sleep(0.002)

    def kernel_name(self):
        return "none (dry run)"

    def episode_stats(self):
        import torch
        ids = torch.arange(self.N, dtype=torch.float32) + self.offset
        return ids, ids.to(torch.int32), torch.ones(self.N, dtype=torch.int8)


# ------------------------------------------------------------------------------------------------ baselines / evidence files
def host_cpu_facts():
    """What the box offers this process: logical CPUs in the affinity mask, physical cores behind them, the cgroup CPU quota."""
    aff = sorted(os.sched_getaffinity(0))
    quota = None
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(p) as f:
                t = f.read().split()
            if p.endswith("cpu.max"):
                quota = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                q = float(t[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota = None if q <= 0 else q / float(f.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    phys = set()
    for c in aff:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "physical_package_id") as f:
                pk = f.read().strip()
            with open(base + "core_id") as f:
                phys.add((pk, f.read().strip()))
        except OSError:
            phys.add(("?", str(c)))
    return {"sched_getaffinity": len(aff), "physical_cores": len(phys), "cgroup_cpu_quota": quota}


def cpu_baseline(n_arenas, level, seed, budget_s=14.0):
    """The oracle (plain-C port of the reference path; one OpenMP team, arena-outer / tick-inner, static blocks of arenas per thread)
    on the host cores: ONE thread, then every thread OpenMP gives the process, in the same run, with the outputs allocated once
    outside the timed loops.  `cores` / `threads` are the threads that did the work (counted inside a parallel region), not the
    affinity mask.  Test infrastructure used only as a reported baseline, never as the measured product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    facts = host_cpu_facts()
    T = 100
    act = None

    def leg(threads, budget):
        nonlocal act
        O.omp_set_threads(threads)
        team = O.omp_team_size()
        w = O.OracleWorld(O.make_config(n_arenas=n_arenas, level=level, seed=seed, auto_reset=True))   # arenas first touched by their threads
        w.reset()
        if act is None:
            act = O.action_tape_uniform(seed, 0, 0, T, n_arenas, w.n_ctrl)   # the first T steps of the same keyed tape the GPU run consumes
        out = w.alloc_rollout_outputs(T)
        w.rollout(act, out=out)   # untimed: first touch of the output pages by the threads that own them, caches warm
        steps, t0 = 0, time.perf_counter()
        while True:
            w.rollout(act, out=out)
            steps += T
            dt = time.perf_counter() - t0
            if dt >= budget:
                break
        return {"threads": team, "value": n_arenas * steps / dt, "ticks": steps, "seconds": dt}

    max_threads = O.omp_max_threads()
    one = leg(1, budget_s * 0.3)
    legs = [one]
    # the thread counts worth timing: what the cgroup quota lets run at once (a 256-CPU box may grant this process 16), one thread per
    # physical core, every logical CPU OpenMP would use by default — each at most once, each capped by the one above it
    phys, quota = facts["physical_cores"], facts["cgroup_cpu_quota"]
    cand = []
    if quota:
        cand.append(max(1, min(max_threads, int(quota + 0.5))))
    cand.append(min(max_threads, phys))
    cand.append(max_threads)
    cand = [t for t in dict.fromkeys(cand) if t > 1]
    if quota:   # more runnable threads than the quota only take turns: timed briefly as evidence, the quota-sized team is the baseline
        cand = cand[:2]
    for i, t in enumerate(cand):
        legs.append(leg(t, budget_s * (0.4 if i == 0 else 0.2)))
    O.omp_set_threads(max_threads)
    best = max(legs, key=lambda r: r["value"])
    for r in legs:
        r["efficiency_vs_one_thread"] = r["value"] / (r["threads"] * one["value"])
    out = {"value": best["value"], "unit": "env-steps/s", "cores": best["threads"], "threads": best["threads"], "kind": "port",
           "one_thread": one["value"], "per_thread": best["value"] / best["threads"], "scaling_efficiency": best["efficiency_vs_one_thread"],
           "legs": legs, "host": facts, "omp_max_threads": max_threads,
           "efficiency_vs_usable_cpus": best["value"] / (min(best["threads"], facts["physical_cores"], facts["cgroup_cpu_quota"] or best["threads"]) * one["value"]),
           "note": "unoptimised scalar C restatement (statement order of the reference, no SIMD): a reported baseline, not a tuned CPU implementation.  "
                   "`value` = the fastest leg; `cores` = `threads` = the OpenMP threads that ran it (counted inside a parallel region); `scaling_efficiency` = "
                   "value / (threads x one_thread); `efficiency_vs_usable_cpus` divides by min(threads, physical cores, cgroup CPU quota) instead: threads beyond the quota "
                   "only take turns, and SMT siblings share a core's FP64 units",
           "sample": f"{n_arenas} arenas x {best['ticks']} ticks in {best['seconds']:.1f} s (one thread: {one['ticks']} ticks in {one['seconds']:.1f} s), same config / seed / "
                     f"keyed action tape (its first {T} steps, cycled), outputs pre-allocated and first-touched outside the timed loop, one OpenMP team, "
                     "arenas outer / ticks inner, schedule(static)"}
    ref = load_json("reference_cpu_rate.json")
    if ref:
        out["reference_python"] = {
            "value": ref["env_steps_per_s"].get(f"lowlevel_2v2_L{level}"), "unit": "env-steps/s", "cores": 1,
            "kind": "the reference's own Python path, measured in the build container (it cannot travel to the GPU box)",
            "where": ref.get("where"), "source": "profiles/reference_cpu_rate.json (oracle/time_reference.py)"}
    return out


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def counter_evidence(instance, n_arenas, ticks, units_per_s_per_gpu, lanes_per_arena):
    """HBM bytes and instruction counts per launch from the PMC passes of a separate rocprofv3 run of this same command
    (tools/prof_pmc.sh -> profiles/latest_traffic.json, profiles/latest_pmc.json; counters cannot be read from inside the
    process).  Stored per arena-tick / per wave-tick, so they apply to any --chunk of the same kernel instance — and ONLY to that
    instance: evidence whose profiled kernel name does not contain `instance` (hh_kernel_instance: the template instance this world
    launches, as a profiler prints it) is stale and is not quoted (traffic / fp64 stay null)."""
    traffic, fp64, issue = None, None, None
    quoted = "builder's rocprofv3 --pmc run of this kernel instance at this arena count (tools/prof_pmc.sh -> profiles/latest_*[_<arenas>].json), NOT measured by this run"
    tj = load_json(f"latest_traffic_{n_arenas}.json") or load_json("latest_traffic.json")
    if tj and tj.get("arenas") == n_arenas and tj.get("hbm_bytes_per_launch") and instance in str(tj.get("kernel_full", tj.get("kernel", ""))):
        per = tj.get("hbm_bytes_per_arena_tick") or tj["hbm_bytes_per_launch"] / (tj["arenas"] * tj["ticks_per_launch"])
        traffic = int(per * n_arenas * ticks)
    pj = load_json(f"latest_pmc_{n_arenas}.json") or load_json("latest_pmc.json")
    if pj and pj.get("arenas") == n_arenas and instance in str(pj.get("kernel", "")):
        valu = pj["insts_valu_per_wave_tick"]
        # every VALU instruction of a wave counted as one issue slot for each lane that carries an aircraft (idle lanes of the
        # 8-arenas-per-wave form are not work)
        lane_ops = valu * lanes_per_arena * units_per_s_per_gpu
        fp64 = {"bound": "fp64 valu issue", "wave_tick": pj.get("wave_tick"), "insts_valu_per_wave_tick": valu, "insts_salu_per_wave_tick": pj.get("insts_salu_per_wave_tick"),
                "achieved_lane_ops_s": lane_ops, "peak": FP64_ISSUE_PEAK, "unit": "lane-ops/s", "frac": lane_ops / FP64_ISSUE_PEAK,
                "frac_note": "every VALU instruction counted as an FP64 issue slot (upper bound on the pipe's use)",
                "source": "profiles/latest_pmc.json: " + quoted}
        f64 = pj.get("insts_f64_per_wave_tick")   # SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64: the instructions that are FP64 arithmetic
        if f64:
            n64 = sum(f64.values())
            fp64["insts_f64_per_wave_tick"] = f64
            fp64["f64_share_of_valu"] = n64 / valu if valu else None
            fp64["true_frac"] = n64 * lanes_per_arena * units_per_s_per_gpu / FP64_ISSUE_PEAK
            fp64["true_frac_note"] = "only ADD/MUL/FMA/TRANS_F64 instructions counted (an FMA as ONE issue slot): the FP64 pipe's arithmetic use"
        # The bound this kernel actually sits on at one wave per SIMD: every instruction of a wave (vector, scalar, LDS) takes one 4-cycle
        # issue slot and nothing overlaps inside a wave (tools/ubench/issue.hip).  slots used = instructions per wave-tick; slots available =
        # the wave-cycles the occupied SIMDs spent on a wave-tick / 4; the rest is waiting (s_waitcnt / barrier) or issue stalls.
        ipw = pj.get("insts_per_wave_tick") or {"VALU": valu, "SALU": pj.get("insts_salu_per_wave_tick", 0), "LDS": pj.get("insts_lds_per_wave_tick", 0)}
        used = float(sum(ipw.values()))
        avail = pj.get("wave_cycles_per_wave_tick", 0) / 4.0
        if avail:
            issue = {"bound": "valu-issue", "slots_used_per_wave_tick": used, "slots_available_per_wave_tick": avail, "frac": used / avail, "unit": "4-cycle issue slots",
                     "insts_per_wave_tick": ipw, "wave_tick": pj.get("wave_tick"),
                     "note": "one instruction per slot, no overlap inside a wave; `frac` = share of the occupied SIMDs' issue slots that carry an instruction "
                             "(simulation + output wave of the two-wave form together)",
                     "source": "profiles/latest_pmc.json: " + quoted}
            q = pj.get("sq_quad_cycles") or {}
            if q.get("SQ_WAVE_CYCLES"):
                wc = float(q["SQ_WAVE_CYCLES"])
                issue["wave_cycles_split"] = {"issuing": q.get("SQ_ACTIVE_INST_ANY", 0) / wc if "SQ_ACTIVE_INST_ANY" in q else None,
                                              "waiting (s_waitcnt, barrier)": q.get("SQ_WAIT_ANY", 0) / wc if "SQ_WAIT_ANY" in q else None,
                                              "issue stalls": q.get("SQ_WAIT_INST_ANY", 0) / wc if "SQ_WAIT_INST_ANY" in q else None}
    return traffic, fp64, issue


def launch_traffic(fname, instance, n_arenas):
    """HBM bytes per launch of a profiled kernel (2*FETCH_SIZE + WRITE_SIZE, tools/prof_pmc.sh) from profiles/<fname>, or None when
    the file was made for another kernel instance or arena count"""
    tj = load_json(fname)
    if tj and tj.get("arenas") == n_arenas and tj.get("hbm_bytes_per_launch") and instance in str(tj.get("kernel_full", "")):
        return int(tj["hbm_bytes_per_launch"])
    return None


def make_tape(args, R, T, N, n_units, step0=0):
    """[T, N, n_units, 4] int8 actions resident in HBM before the timed region.  keyed (default): i.i.d. uniform over
    MultiDiscrete([13,9,2,2]) from the keyed RNG, key = (seed, GLOBAL arena, step, agent) — rank r's tape is the slice of the one global
    tape its arenas select (hh_action_tape_uniform, pinned against the oracle's in tests/test_gpu_parity.py)"""
    torch = R.torch
    if args.tape == "keyed":
        from hhmarl_2d_amd.world import action_tape_uniform
        return action_tape_uniform(args.seed, R.rank * N, step0, T, N, n_units, device=R.dev)
    gen = torch.Generator(device=R.dev)
    gen.manual_seed(args.seed + R.rank + 7919 * step0)
    hi = torch.tensor([13, 9, 2, 2], device=R.dev)
    return (torch.rand((T, N, n_units, 4), device=R.dev, generator=gen) * hi).to(torch.int8).contiguous()


TAPE_DESC = {"keyed": "actions i.i.d. uniform over MultiDiscrete([13,9,2,2]) from the keyed RNG, key = (seed, global arena, step, agent)",
             "torch": "actions i.i.d. uniform over MultiDiscrete([13,9,2,2]) from torch.rand (per-rank seed)"}


# ------------------------------------------------------------------------------------------------ configs[1]: the headline
def main_low(args, R=None):
    own = R is None
    R = R or Ranks(args)
    torch = R.torch
    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas or 4096
    chunk = max(1, args.chunk)
    sw = ShardedWorld(dict(n_arenas=N, level=args.level, seed=args.seed, auto_reset=True), rank=R.rank, world_size=R.world,
                      device=R.local_rank, world_factory=DryWorld if R.dry else None)
    w = sw.world
    w.reset()
    tape, out, side = None, w.alloc_outputs(chunk), None
    n_tape = 4 if N * chunk <= (1 << 26) else 2  # distinct chunks of actions, cycled (8 B per arena-tick: 1 GB per chunk at 262144 arenas x 500 ticks)
    if not R.dry:
        # action tape resident in HBM before the timed region: i.i.d. uniform MultiDiscrete([13,9,2,2])
        tape = make_tape(args, R, n_tape * chunk, N, w.n_ctrl).view(n_tape, chunk, N, w.n_ctrl, 4)
        side = torch.cuda.Stream()
    state = {"k": 0}

    def launch(timed_events=None):
        k = state["k"]
        if timed_events is not None and not R.dry:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        w.rollout(None if R.dry else tape[k % n_tape], out=out)
        if timed_events is not None and not R.dry:
            e1.record()
            timed_events.append((e0, e1))
        state["k"] = k + 1
        if state["k"] % args.log_every == 0:
            sw.log_episode_stats(side)   # logging only: snapshot = one small launch, RCCL all-gather on the side stream

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup and not R.dry:   # clocks ramp up from idle: not part of W or K
        launch()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        launch()
    R.barrier()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch(evs)
    R.barrier()
    dt_local = time.perf_counter() - t0
    dt = R.max_over_ranks(dt_local)
    sw.log_episode_stats(side)
    R.barrier()
    per_rank = R.gather_floats(N * chunk * args.steps / dt_local)

    value = N * chunk * args.steps * R.world / dt
    line = {
        "metric": "env-steps/sec (2v2)", "value": value, "unit": "env-steps/s", "n_gpus": R.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "agent_steps_per_s": value * 2, "per_rank_env_steps_per_s": per_rank,
        "config": {"workload": f"{N} arenas/GPU x 2-vs-2 fight L{args.level} (scripted opponent), random actions, auto-reset "
                               f"(BASELINE configs[1])", "arenas_per_gpu": N, "ticks_per_step": chunk,
                   "actions": TAPE_DESC[args.tape] + f"; {n_tape} chunks of {chunk} steps resident in HBM, cycled",
                   "timed_ticks_per_arena": chunk * args.steps, "warmup_ticks_per_arena": chunk * args.warmup,
                   "env_steps_per_step": N * chunk * R.world,
                   "step": "one hh_rollout launch = ticks_per_step consecutive LowLevelEnv.step() calls of every arena",
                   "parallelism": f"arena-sharded x{R.world}, no data-path collective; logging all-gather every {args.log_every} launches on a side stream"},
    }
    if R.world > 1 and args.one_gpu_value:
        line["scaling_efficiency"] = value / (R.world * args.one_gpu_value)
        line["one_gpu_value"] = args.one_gpu_value
    # did the logging collective see every rank?  (rank, first global arena, arenas) of each rank travel over the same backend as the statistics
    line.update(sw.evidence())
    if R.dry:
        line["dry_run"] = True
    else:
        # dominant kernel: average launch duration from HIP events on the launch stream
        durs = [a.elapsed_time(b) * 1e-3 for a, b in evs]
        avg_launch_s = sum(durs) / len(durs)
        bytes_per_launch = ALGO_BYTES_2V2_STEP * N * chunk
        achieved = bytes_per_launch / avg_launch_s / 1e9
        kname = w.kernel_name()
        traffic, fp64, issue = counter_evidence(w.kernel_instance(), N, chunk, N * chunk / avg_launch_s, 4)   # 2-vs-2: four aircraft lanes per arena
        line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic, "kernel": kname, "kernel_instance": w.kernel_instance(), "avg_launch_ms": avg_launch_s * 1e3,
                            "launch_ms_min": min(durs) * 1e3, "launch_ms_max": max(durs) * 1e3, "launches_timed": len(durs),
                            "traffic_source": None if traffic is None else "profiles/latest_traffic.json: builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                                                           "kernel instance at this arena count (tools/prof_pmc.sh), NOT measured by this run",
                            "algorithmic_bytes_per_launch": bytes_per_launch, "issue": issue, "fp64": fp64,
                            "note": "instruction-issue bound, not HBM bound (DESIGN.md section 4): `issue.frac` is the share of the occupied SIMDs' issue "
                                    "slots that carry an instruction, `fp64` the share of the vector-FP64 pipe; HBM traffic is far below the algorithmic "
                                    "bytes because state stays in registers across the ticks of a launch.  `achieved` / `frac` are measured by this run "
                                    "(HIP events); `traffic`, `issue` and `fp64` are quoted from the committed counter passes"}
        if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, args.level, args.seed)
    if not own:
        return line
    if R.rank == 0:
        print(json.dumps(line), flush=True)
    R.close()


# ------------------------------------------------------------------------------------------------ configs[2]: policy in the loop
def main_policy_rollout(args, R=None):
    """BASELINE configs[2]: N arenas x 2-vs-2 fight L3 driven by a policy in the loop — per tick: observations [N, 2, 26|24]
    -> fight networks with the reference's architecture (models/ac_models_hetero.py Fight1 for agent 1, Fight2 for agent 2, actor
    half, greedy decode as env_base.py:373-382) in the fused HIP kernel -> int8 actions -> hh_step.  A step is one tick of all arenas."""
    own = R is None
    R = R or Ranks(args)
    torch = R.torch
    from hhmarl_2d_amd.pilots import SEL_FIGHT1, SEL_FIGHT2, PolicyBank
    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas or 16384
    K = args.streams or DEFAULT_STREAMS["rollout"]
    assert N % K == 0, "--streams must divide the arena count"
    n = N // K
    if K > 1 and "HH_FORCE_W" not in os.environ and not args.joined and not args.no_graph:
        # sub-worlds side by side: the single-wave world kernel at 256 registers (two waves per SIMD) and the 64-row policy tiles (half a CU) can share CUs, which the
        # two-wave world kernel (313 registers a wave) and 128-row tiles cannot: 16384 arenas, K = 2: 2.21e8 against 1.96e8 env-steps/s (read at hh_world_create)
        os.environ["HH_FORCE_W"] = "2"
    # sub-world k of rank r holds the global arenas [r N + k n, r N + (k + 1) n): the same arenas as one world of N (keyed RNG by global id)
    sws = [ShardedWorld(dict(n_arenas=n, level=args.level, seed=args.seed, auto_reset=True, arena_offset=k * n + R.rank * (N - n)), rank=R.rank,
                        world_size=R.world, device=R.local_rank) for k in range(K)]
    sw, w = sws[0], sws[0].world
    ppo = bool(getattr(args, "ppo", False))
    banks = [(PolicyBank.trainable_init(R.dev, seed=args.seed, max_rows=n * 2) if ppo else PolicyBank.random_init(R.dev, seed=args.seed, max_rows=n * 2)) for _ in range(K)]
    bank = banks[0]
    outs, acts = [], []
    logps = [torch.zeros((n, 2), dtype=torch.float32, device=R.dev) for _ in range(K)]
    vfs = [torch.zeros((n, 2), dtype=torch.float32, device=R.dev) for _ in range(K)]

    def policy(k, first=False):
        """one policy evaluation of sub-world k's agents: greedy actor (frozen-policy form) or the PPO sampler's actor + draw + logp + value"""
        sel = net_id if first else None
        if ppo:
            banks[k].sample(outs[k][0], sel, world=sws[k].world, actions=acts[k], logp=logps[k], vf=vfs[k])
        else:
            banks[k].act(outs[k][0], sel, acts[k])
    # agent 1 is a type-1 aircraft (Fight1), agent 2 a type-2 (Fight2): env_base.py:560-561 fixes the first two slots
    net_id = torch.tensor([SEL_FIGHT1, SEL_FIGHT2], dtype=torch.uint8, device=R.dev).repeat(n, 1).contiguous()
    for k in range(K):
        o = sws[k].world.alloc_outputs()
        o[0].copy_(sws[k].world.reset())
        outs.append(o)
        acts.append(torch.zeros((n, 2, 4), dtype=torch.int8, device=R.dev))
        policy(k, first=True)                   # builds the row lists once: agent 1 -> Fight1, agent 2 -> Fight2 never changes
    out, act = outs[0], acts[0]
    streams = [torch.cuda.Stream() for _ in range(K)] if K > 1 else []

    def tick():
        if K == 1:
            policy(0)                     # same selectors as before: no binning pass
            w.step(act, out=out)
            return
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                policy(k)
                sws[k].world.step(acts[k], out=outs[k])
        for k in range(K):
            cur.wait_stream(streams[k])

    graph = None
    pipelined = K > 1 and not args.no_graph and not args.joined   # one graph per sub-world on its own stream, no join between ticks (main_hier_split)
    if pipelined:
        streams = make_streams(torch, K)
        for _ in range(3):
            tick()
        torch.cuda.synchronize()
        graphs = []
        for k in range(K):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, stream=(streams[k] if streams[k].cuda_stream != 0 else torch.cuda.Stream())):
                policy(k)
                sws[k].world.step(acts[k], out=outs[k])
            graphs.append(gk)

        def run():
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
    elif not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                tick()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tick()
    if not pipelined:
        run = graph.replay if graph is not None else tick
    log_side = torch.cuda.Stream() if not pipelined else None

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        run()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run()
    R.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    if pipelined:
        for k in range(K):
            streams[k].wait_stream(torch.cuda.current_stream())
    for k in range(args.steps):
        run()
        if k % 256 == 255 and log_side is not None:
            sw.log_episode_stats(log_side)
    if pipelined:
        for k in range(K):
            torch.cuda.current_stream().wait_stream(streams[k])
    e1.record()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    gpu_s = e0.elapsed_time(e1) * 1e-3
    value = N * R.world * args.steps / dt
    achieved = ALGO_BYTES_2V2_STEP * N * args.steps / gpu_s / 1e9
    line = {
        "metric": "env-steps/sec (2v2, PPO sampler in the loop)" if ppo else "env-steps/sec (2v2, policy in the loop)", "value": value, "unit": "env-steps/s", "n_gpus": R.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 world + f32 policy", "data": "synthetic", "agent_steps_per_s": value * 2, "streams": K,
        "config": {"workload": (f"{N} arenas/GPU x 2-vs-2 fight L{args.level} driving a PPO fight-policy rollout: every tick random-init Fight1/Fight2 "
                                f"(reference architecture) are evaluated the way RLlib's sampler does (train_hetero.py:206-243) — actor logits, a Categorical "
                                f"draw per action component from the keyed RNG, its log-probability, and the centralised value branch on the other agent's "
                                f"observation — in the fused HIP kernel on the same GPU, then hh_step; auto-reset (BASELINE configs[2])") if ppo else
                               (f"{N} arenas/GPU x 2-vs-2 fight L{args.level}, actions from random-init Fight1/Fight2 actors (reference "
                                f"architecture, fp32, fused HIP kernel, greedy decode) evaluated every tick on the same GPU, auto-reset "
                                f"(BASELINE configs[2] with a frozen / evaluation policy: no draw, no logp, no value)"),
                   "arenas_per_gpu": N, "ticks_per_step": 1, "parallelism": f"arena-sharded x{R.world}, no data-path collective" + (f"; {K} sub-worlds, one HIP graph per sub-world on its own stream" if pipelined else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": f"{w.kernel_name()} (T = 1 per launch) + hh_k_policy, " + ("one HIP graph per tick" if graph is not None else "eager"),
                     "policy_flops_per_s": bank.flops_per_row(PolicyBank.FIGHT1) * N * 2 * args.steps / gpu_s},
    }
    # the two kernels of a tick on their own (HIP events, eager launches): which one dominates and its own roofline
    def timed(fn, n=60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n
    from hhmarl_2d_amd import policy_nets as PN
    pol_ms = timed(lambda: policy(0))
    world_ms = timed(lambda: w.step(act, out=out))
    useful = (bank.flops_per_row(PolicyBank.FIGHT1) + bank.flops_per_row(PolicyBank.FIGHT2)) * n     # one sub-world's rows, fp32-equivalent
    if ppo:
        useful += (PN.critic_flops_per_row(PN.FIGHT1) + PN.critic_flops_per_row(PN.FIGHT2)) * n      # the value branch of every row
    pname = bank.kernel_name(2 * n, sampler=ppo)
    line["kernels_ms"] = {pname: pol_ms, w.kernel_instance(): world_ms, "arenas_per_launch": n}
    issued = 3.0 * useful   # split-fp16: hi*hi + hi*lo + lo*hi = three MFMA passes per product
    peak = MFMA_F16_PEAK_TFLOPS
    line["roofline"]["dominant"] = {
        "kernel": pname + " (split-fp16: hi*hi + hi*lo + lo*hi on " + ("v_mfma_f32_16x16x32_f16)" if "w16" in pname else "v_mfma_f32_32x32x16_f16)"), "bound": "mfma", "avg_launch_ms": pol_ms,
        "achieved": useful / (pol_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": useful / (pol_ms * 1e-3) / 1e12 / peak,
        "frac_note": "USEFUL flops (one fp32-equivalent multiply-add per weight and row) over the dense fp16 MFMA peak; the three emulation passes "
                     "that buy fp32 accuracy are matrix-pipe work, not useful work: `issued_frac` counts them",
        "issued_tflops": issued / (pol_ms * 1e-3) / 1e12, "issued_frac": issued / (pol_ms * 1e-3) / 1e12 / peak}
    line["roofline"]["dominant"]["traffic"] = None
    for tf in ("latest_policy_traffic.json", "latest_policy_ppo_traffic.json"):   # the greedy call's instance | the sampler's
        tr = load_json(tf)
        if tr and tr.get("rows") == 2 * n and pname in str(tr.get("kernel_full", "")):
            line["roofline"]["dominant"]["traffic"] = int(tr["hbm_bytes_per_launch"])
            line["roofline"]["dominant"]["traffic_source"] = f"profiles/{tf} (builder's rocprofv3 --pmc run of this kernel instance, not this run)"
    if not own:
        return line
    if R.rank == 0:
        print(json.dumps(line), flush=True)
    R.close()


def main_collect(args, R=None):
    """BASELINE configs[2] as a whole PPO batch: one step = one `PPORollout.collect` of T = --chunk ticks of N arenas — per tick the sampler
    (actor, keyed Categorical draw, logp, centralised value branch: hh_policy_sample) and hh_step, then the bootstrap value evaluation and
    hh_gae_rllib over the [T, N, 2] buffers (RLlib's trajectory semantics, rollout.py) — replayed from ONE HIP graph; everything stays on
    the device.  value = N x T x steps / wall time."""
    own = R is None
    R = R or Ranks(args)
    torch = R.torch
    from hhmarl_2d_amd.pilots import PolicyBank
    from hhmarl_2d_amd.rollout import PPORollout
    from hhmarl_2d_amd.sharding import ShardedWorld
    N, T = args.arenas or 16384, max(1, args.chunk)
    sw = ShardedWorld(dict(n_arenas=N, level=args.level, seed=args.seed, auto_reset=True), rank=R.rank, world_size=R.world, device=R.local_rank)
    w = sw.world
    bank = PolicyBank.trainable_init(R.dev, seed=args.seed, max_rows=2 * N)
    ro = PPORollout(w, bank, T, use_graph=not args.no_graph)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        ro.collect()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        ro.collect()
    R.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        ro.collect()
    e1.record()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    gpu_s = e0.elapsed_time(e1) * 1e-3
    value = N * T * R.world * args.steps / dt
    comp = float(ro.complete.float().mean())
    achieved = ALGO_BYTES_2V2_STEP * N * T * args.steps / gpu_s / 1e9
    line = {
        "metric": "env-steps/sec (2v2, whole PPO batch: sampler + step per tick, bootstrap, GAE)", "value": value, "unit": "env-steps/s", "n_gpus": R.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 world + f32 policy, f64 GAE recursion", "data": "synthetic", "agent_steps_per_s": value * 2, "gpu_ms_per_step": gpu_s / args.steps * 1e3,
        "collect": {"ticks": T, "launches": 2 * T + 2, "semantics": ro.semantics, "gae": "hh_gae_rllib (gamma 0.99, lambda 0.95)",
                    "rows_in_complete_episodes": comp, "buffers": "obs, actions, logp, vf, reward, valid, done, adv, target of T x N x 2 rows, resident in HBM"},
        "config": {"workload": f"{N} arenas/GPU x 2-vs-2 fight L{args.level}: PPORollout.collect of {T} ticks = what RLlib's rollout workers hand train_hetero.py's learner "
                               f"(train_hetero.py:206-243: sampler every tick, complete-episode mask, GAE gamma 0.99 / lambda 0.95), random-init Fight1/Fight2 with ONE tied "
                               f"shared layer, one HIP graph per collect, auto-reset (BASELINE configs[2])",
                   "arenas_per_gpu": N, "ticks_per_step": T, "parallelism": f"arena-sharded x{R.world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": f"{bank.kernel_name(2 * N, sampler=True)} + {w.kernel_instance()} (T = 1 per launch) + hh_k_gae_rllib, one HIP graph per collect",
                     "note": "1 113 algorithmic bytes per env step (SURVEY.md 8d); the batch is policy-bound: see extra.configs2.roofline.dominant for the sampler kernel's MFMA fraction"},
    }
    bank.close()
    if not own:
        return line
    if R.rank == 0:
        print(json.dumps(line), flush=True)
    R.close()


# ------------------------------------------------------------------------------------------------ configs[3]/[4]: HighLevelEnv
PILOT_DESC = {"tape": "uniform action tape resident in HBM", "random": "uniform actions drawn by torch kernels inside the step",
              "mlp": "random-init MLP stand-ins (torch)", "net": "random-init Fight1/Fight2/Esc1/Esc2 actors (reference architecture) in the fused HIP kernel"}


def commander_tape(args, R, N):
    """64 commander steps of uniform {0,1,2} actions [64, N, 3] int8: the speed component (uniform over 0..8) of the keyed action word mod 3, at step
    indices far from the pilots' tape (or torch.rand with --tape torch)"""
    return (make_tape(args, R, 64, N, 3, step0=1 << 20)[..., 1] % 3).to(R.torch.int8).contiguous()


def main_hier(args, R=None):
    """BASELINE configs[3]/[4]: N arenas x 3-vs-3 HighLevelEnv (map 0.5, horizon 500, N_OPPS_HL=2), commander actions uniform
    {0,1,2}; a step is one commander step (HighLevelEnv.step) of every arena = up to 16 sub-steps with pilot actions."""
    own = R is None
    R = R or Ranks(args)
    torch = R.torch
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.pilots import MLPPilot, NetPilot, RandomPilot, TapePilot
    from hhmarl_2d_amd.sharding import ShardedWorld
    N = args.arenas or 8192
    K = (args.streams or DEFAULT_STREAMS["hier_net_variants" if args.pilot_rows == "variants" else "hier_net"]) if args.pilot == "net" else 1
    if K > 1 or args.pilot == "net":   # the networks-in-the-loop workload always runs the sub-world form (K = 1: one sub-world)
        return main_hier_split(args, R, own, N, K)
    sw = ShardedWorld(dict(n_arenas=N, env_kind=1, seed=args.seed, auto_reset=True), rank=R.rank, world_size=R.world, device=R.local_rank)
    w = sw.world
    w.reset()
    if args.pilot == "tape":
        pilot = TapePilot(R.dev, N, 6, seed=args.seed + R.rank, bank=make_tape(args, R, 4 * 16, N, 6).view(4, 16, N, 6, 4))
    elif args.pilot == "random":
        pilot = RandomPilot(R.dev, args.seed + R.rank)
    elif args.pilot == "net":
        pilot = NetPilot(w, seed=args.seed, bind=os.environ.get("HH_BENCH_NO_BIND", "0") != "1")   # HH_BENCH_NO_BIND=1: binning pass per call (A/B)
    else:
        pilot = MLPPilot(R.dev, seed=args.seed)
    cmds = commander_tape(args, R, N)
    out, pbuf = w.alloc_outputs(), w.alloc_pilot()

    # pilot networks in the loop: the macro step is a fixed sequence of world launches + the pilots' kernels: capture it once
    # into a HIP graph (launch-bound inner loop; no host synchronisation inside) and replay it per commander step.
    # actions from a resident tape: the whole commander step is ONE persistent launch (hh_hl_rollout).
    cmd_static = cmds[0].clone()
    graph = None
    one_launch = args.pilot == "tape" and not args.phases
    if not args.no_graph and not one_launch:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                macro_step(w, cmd_static, pilot, out=out, pilot_buf=pbuf)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            macro_step(w, cmd_static, pilot, out=out, pilot_buf=pbuf)
    log_side = torch.cuda.Stream()
    state = {"k": 0}

    def run(n):
        for _ in range(n):
            k = state["k"]
            if one_launch:
                w.hl_rollout(cmds[k % 64], pilot.bank[k % pilot.bank.shape[0]], out=out)
                state["k"] = k + 1
                if state["k"] % 16 == 0:
                    sw.log_episode_stats(log_side)
                continue
            if args.pilot == "tape":
                pilot.load(k)
            if graph is not None:
                cmd_static.copy_(cmds[k % 64])
                graph.replay()
            else:
                macro_step(w, cmds[k % 64], pilot, out=out, pilot_buf=pbuf)
            state["k"] = k + 1
            if state["k"] % 16 == 0:
                sw.log_episode_stats(log_side)

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        run(1)
        torch.cuda.synchronize()
    run(args.warmup)
    R.barrier()
    ticks0 = w.hl_tick_count()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run(args.steps)
    e1.record()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    ticks = w.hl_tick_count() - ticks0            # arena-ticks actually run (arenas leave a macro step early)
    gpu_s = e0.elapsed_time(e1) * 1e-3
    steps = args.steps
    value = N * R.world * steps / dt
    tape_launch = one_launch
    tick_bytes = ALGO_BYTES_3V3_TICK_TAPE if tape_launch else ALGO_BYTES_3V3_TICK
    algo_bytes = tick_bytes * ticks + ALGO_BYTES_3V3_CMD_FIXED * N * steps
    achieved = algo_bytes / gpu_s / 1e9
    per_rank = R.gather_floats(N * steps / dt)
    line = {
        "metric": "commander-steps/sec (3v3 HighLevelEnv)", "value": value, "unit": "env-steps/s", "n_gpus": R.world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "agent_steps_per_s": value * 3, "per_rank_commander_steps_per_s": per_rank,
        "sim_ticks_per_s": ticks * R.world / dt, "ticks_per_commander_step": ticks / float(N * steps),
        "config": {"workload": f"{N} arenas/GPU x 3-vs-3 HighLevelEnv commander steps (<= 16 sub-steps each), uniform commander actions, "
                               f"pilots = {PILOT_DESC[args.pilot]}, auto-reset (BASELINE configs[{3 if R.world == 1 else 4}])", "arenas_per_gpu": N,
                   "parallelism": f"arena-sharded x{R.world}, no data-path collective; logging all-gather of [N, 3] episode statistics every 16 commander steps on a side stream"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": (f"{w.kernel_instance(1)} (one persistent launch per commander step)" if one_launch else
                                                 f"{w.kernel_name()} (every phase launch of the macro step, plus the pilots' kernels if any)"),
                     "algorithmic_bytes": algo_bytes,
                     "note": (f"{tick_bytes} B per arena-tick actually run + 424 B per commander step (SURVEY.md 8d" +
                              ("; the 720 B of pilot observations per tick are not charged: with the actions on a tape nobody builds them)" if tape_launch else ")"))},
    }
    if tape_launch:   # what the fused schedule has to move: the state once per commander step, the pilots' action words per tick, the commander's rows
        sched = (2 * ALGO_BYTES_3V3_STATE + ALGO_BYTES_3V3_CMD_FIXED) * N * steps + 24 * ticks
        line["roofline"]["schedule"] = {"bytes": sched, "achieved": sched / gpu_s / 1e9, "frac": sched / gpu_s / 1e9 / HBM_PEAK_GBS,
                                        "note": "bytes the one-launch schedule needs (state stays in registers across the sub-steps): the figure the counter traffic should be compared with"}
    line["gpu_ms_per_step"] = gpu_s / steps * 1e3
    if log_side is not None:
        sw.log_episode_stats(log_side)
        line.update(sw.evidence())   # ranks_seen, block order and gathered rows of the logging collective (all ranks call it)
    if tape_launch:   # HBM bytes per commander step of the one-launch kernel, from the committed PMC passes
        tf = "latest_hier_traffic.json" if load_json("latest_hier_traffic.json") else "r03_hier8192_traffic.json"
        tr = launch_traffic(tf, w.kernel_instance(1), N)
        line["roofline"]["traffic"] = tr
        if tr is not None:
            line["roofline"]["traffic_frac"] = tr / (gpu_s / steps) / 1e9 / HBM_PEAK_GBS
            line["roofline"]["traffic_source"] = f"profiles/{tf}: builder's rocprofv3 --pmc passes of this kernel instance (bytes per commander step), NOT measured by this run"
    if not one_launch:
        line["launches_per_step"] = 2 + 16 * (4 if args.pilot in ("net", "mlp", "random") else 2)
    if hasattr(pilot, "close"):
        pilot.close()
    if not own:
        return line
    if R.rank == 0:
        print(json.dumps(line), flush=True)
    R.close()


def make_streams(torch, K, allow_default=True):
    """K streams on K different hardware queues.  The runtime drives FOUR hardware queues (more, GPU_MAX_HW_QUEUES, do not run side by side: tools/queue_probe.py);
    the default stream owns one of them and every other stream shares the remaining three, so four dependent chains only run side by side when one of them is
    issued on the default stream (tools/timeline.py: with four side streams, sub-world 3 starts when one of the others has finished its commander step).
    torch.cuda.Stream() also hands out its pool low / high priority in turn — consecutive pool streams land on two queues — and a stream that ever carried work
    (a warm-up side stream) keeps its queue: the streams are created directly, one after the other.  HH_BENCH_STREAMS = side (no default stream) | pool: A/B."""
    mode = os.environ.get("HH_BENCH_STREAMS", "default0")
    if mode == "pool":
        return [torch.cuda.Stream() for _ in range(K)]
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = []
    if mode == "default0" and K > 3 and allow_default:
        out.append(torch.cuda.default_stream())
    for _ in range(K - len(out)):
        h = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithFlags(ctypes.byref(h), ctypes.c_uint(1))   # hipStreamNonBlocking
        if rc != 0:
            raise RuntimeError(f"hipStreamCreateWithFlags: {rc}")
        out.append(torch.cuda.ExternalStream(h.value))
    return out


def main_hier_split(args, R, own, N, K):
    """configs[3] with the pilot networks in the loop, the arenas split into K sub-worlds (disjoint global arena ids: the same arenas
    as one world of N) whose 66-launch commander steps run on K streams inside ONE HIP graph: a sub-world's world-phase launches
    (single-pass, latency-bound) run under another sub-world's policy kernel"""
    torch = R.torch
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.pilots import NetPilot, VariantNetPilot
    from hhmarl_2d_amd.sharding import ShardedWorld
    sizes = [N // K + (1 if k < N % K else 0) for k in range(K)]   # K need not divide N: the sub-worlds tile [0, N) with sizes differing by at most one arena
    offs = [sum(sizes[:k]) for k in range(K)]
    n = sizes[0]
    variants = args.pilot_rows == "variants"
    if variants and "HH_POLICY_W" not in os.environ and "HH_POLICY_TILE" not in os.environ and K > 1:
        # calls of ~10 k listed rows from several streams at once: the streamed form with 128-row tiles is ahead of what the bank's back-to-back heuristic
        # picks (4.64e6 against 3.92e6 commander-steps/s at K = 4; tools/variants_rates2.sh); read by hh_policy_create
        os.environ["HH_POLICY_W"] = "3"
    # ShardedWorld adds rank * n_arenas to the offset it is given: sub-world k of rank r starts at global arena r N + offs[k]
    sws = [ShardedWorld(dict(n_arenas=sizes[k], env_kind=1, seed=args.seed, auto_reset=True, arena_offset=offs[k] + R.rank * (N - sizes[k])), rank=R.rank, world_size=R.world,
                        device=R.local_rank) for k in range(K)]
    worlds = [x.world for x in sws]
    for w in worlds:
        w.reset()
    pilots_ = [(VariantNetPilot if variants else NetPilot)(w, seed=args.seed) for w in worlds]
    if "HH_POLICY_TILE" not in os.environ and (n * 3 <= 10240 if not variants else pilots_[0].live <= 10240):
        for pl in pilots_:   # small calls on concurrent streams stay on the tile forms: wide tiles, the other streams fill what a partial round leaves idle
            pl.bank.set_tile_rows(64)
    cmds = commander_tape(args, R, N)
    cmd_static = [cmds[0, offs[k]:offs[k] + sizes[k]].clone() for k in range(K)]
    outs = [w.alloc_outputs() for w in worlds]
    pbufs = [(w.alloc_pilot_variants() if variants else w.alloc_pilot()) for w in worlds]
    pipelined = not args.no_graph and not args.joined
    streams = make_streams(torch, K, allow_default=pipelined)   # (a joined graph is captured: nothing may be issued on the default stream meanwhile)

    def step():
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                macro_step(worlds[k], cmd_static[k], pilots_[k], out=outs[k], pilot_buf=pbufs[k])
        for k in range(K):
            cur.wait_stream(streams[k])

    for _ in range(2):   # (no side stream here: every stream that ever carried work keeps one of the four hardware queues, and K of them are needed below)
        step()
    torch.cuda.synchronize()
    # One graph per sub-world, replayed on the sub-world's own stream (default), or ONE graph whose K branches join at the end of every commander step
    # (--joined).  hipGraphLaunch enqueues a graph's kernel nodes branch after branch at ~3 us of host time each: branch k of the joined graph starts
    # k x 100 us after branch 0 and the step ends with the last branch alone on the chip (tools/timeline.py).  Separate graphs let every sub-world go on
    # with its next commander step as soon as its own last launch is through — the arenas are independent environments, nothing orders sub-world 0's step
    # n + 1 after sub-world 3's step n — and the host only has to keep K queues fed (34 launches per sub-world and step).  The timed region still covers
    # exactly `steps` commander steps of every arena, bracketed by a device synchronisation on both sides.
    graph, graphs = None, []
    if pipelined:
        for k in range(K):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, stream=(streams[k] if streams[k].cuda_stream != 0 else torch.cuda.Stream())):
                macro_step(worlds[k], cmd_static[k], pilots_[k], out=outs[k], pilot_buf=pbufs[k])
            graphs.append(gk)
    elif not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
    state = {"k": 0}

    def run(m):
        cur = torch.cuda.current_stream()
        if pipelined:
            for j in range(K):
                streams[j].wait_stream(cur)
        for _ in range(m):
            k = state["k"]
            if pipelined:
                for j in range(K):
                    with torch.cuda.stream(streams[j]):
                        cmd_static[j].copy_(cmds[k % 64, offs[j]:offs[j] + sizes[j]])
                        graphs[j].replay()
            else:
                for j in range(K):
                    cmd_static[j].copy_(cmds[k % 64, offs[j]:offs[j] + sizes[j]])
                if graph is not None:
                    graph.replay()
                else:   # --no-graph: the same launches issued eagerly on the K streams (A/B of the graph's scheduling)
                    step()
            state["k"] = k + 1
        if pipelined:
            for j in range(K):
                cur.wait_stream(streams[j])

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:
        run(1)
        torch.cuda.synchronize()
    run(args.warmup)
    R.barrier()
    ticks0 = sum(w.hl_tick_count() for w in worlds)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run(args.steps)
    e1.record()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    ticks = sum(w.hl_tick_count() for w in worlds) - ticks0
    gpu_s = e0.elapsed_time(e1) * 1e-3
    steps = args.steps
    value = N * R.world * steps / dt
    algo_bytes = ALGO_BYTES_3V3_TICK * ticks + ALGO_BYTES_3V3_CMD_FIXED * N * steps
    achieved = algo_bytes / gpu_s / 1e9
    line = {
        "metric": "commander-steps/sec (3v3 HighLevelEnv)", "value": value, "unit": "env-steps/s", "n_gpus": R.world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "agent_steps_per_s": value * 3, "streams": K,
        "sim_ticks_per_s": ticks * R.world / dt, "ticks_per_commander_step": ticks / float(N * steps),
        "config": {"workload": f"{N} arenas/GPU x 3-vs-3 HighLevelEnv commander steps (<= 16 sub-steps each), uniform commander actions, "
                               f"pilots = {PILOT_DESC['net']}, auto-reset (BASELINE configs[3])", "arenas_per_gpu": N,
                   "parallelism": f"arena-sharded x{R.world}, no data-path collective; {K} sub-worlds on {K} streams, " + ("one HIP graph per sub-world (no join between commander steps)" if pipelined else "inside one HIP graph (joined at every commander step)"),
                   "pilot_rows": ("variants: one launch + one policy call per sub-step (hh_hl_begin_variants / hh_hl_act_tick)" if variants else
                                  "sides: a launch and a policy call per side and sub-step")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": (f"hh_k_hier_oct_v (one launch per sub-step) + {'hh_k_policy_w16<8>' if os.environ.get('HH_POLICY_W') == '3' else 'hh_k_policy_* by row count'}" if variants else
                                f"{worlds[0].kernel_instance(0)} (every phase launch of the macro step) + hh_k_policy_h"), "algorithmic_bytes": algo_bytes,
                     "note": "2056 B per arena-tick actually run + 424 B per commander step (SURVEY.md 8d)"},
        "gpu_ms_per_step": gpu_s / steps * 1e3, "launches_per_step": (34 if variants else 66) * K, "launches_per_sub_world_step": 34 if variants else 66,
    }
    for p in pilots_:
        p.close()
    if not own:
        return line
    if R.rank == 0:
        print(json.dumps(line), flush=True)
    R.close()


def main():
    args = parse_args()
    if args.workload == "collect" and args.chunk == 500:
        args.chunk = 64     # T of a collect unless given
    if args.dry_run and args.workload != "low":
        sys.exit("--dry-run exercises the rank plumbing of the default workload on a CPU box (gloo); the other workloads need the GPU")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.workload == "hier":
        return main_hier(args)
    if args.workload == "rollout":
        return main_policy_rollout(args)
    if args.workload == "collect":
        return main_collect(args)
    single = args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
    if not single and not args.no_extra:
        # N ranks: configs[1] per rank stays the headline `value`; BASELINE configs[4] (8192 arenas x 3-vs-3 HighLevelEnv per rank, the episode
        # statistics all-gathered over RCCL on a side stream) rides along, measured by the same ranks right after it
        R = Ranks(args)
        line = main_low(args, R)
        extra = configs4(args, R)
        if R.rank == 0:
            line["extra"] = {"configs4": extra}
            print(json.dumps(line), flush=True)
        R.close()
        return
    if args.dry_run or args.no_extra:
        return main_low(args)
    # the driver's line: configs[1] is the headline; the other single-GPU configurations of BASELINE.json ride along as short runs
    R = Ranks(args)
    line = main_low(args, R)
    line["extra"] = extra_configs(args, R)
    print(json.dumps(line), flush=True)
    R.close()


def configs4(args, R):
    """BASELINE configs[4] on the ranks of this job: 8192 arenas/rank x 3-vs-3 HighLevelEnv, pilot actions from a resident tape (one persistent
    launch per commander step), logging all-gather on a side stream.  --dry-run: the rank plumbing and the gather only (gloo, no kernel)."""
    import copy
    N4 = 8192
    if R.dry:
        from hhmarl_2d_amd.sharding import ShardedWorld
        sw = ShardedWorld(dict(n_arenas=N4, env_kind=1, seed=args.seed, auto_reset=True), rank=R.rank, world_size=R.world, device=R.local_rank, world_factory=DryWorld)
        R.barrier()
        sw.log_episode_stats(None)
        R.barrier()
        first = None if sw.last_stats is None else [float(sw.last_stats[i * N4, 0]) for i in range(R.world)]   # DryWorld reports the global arena id: ranks in order
        ev = sw.evidence()
        assert first is None or first == [float(x) for x in ev["first_global_arena_of_each_block"]], (first, ev)   # the statistics blocks sit where the ranks say they are
        return dict({"dry_run": True, "arenas_per_gpu": N4, "n_gpus": R.world}, **ev)
    a = copy.copy(args)
    a.workload, a.pilot, a.arenas, a.steps, a.warmup, a.spinup, a.phases, a.streams = "hier", "tape", N4, 30, 6, 0.3, False, 0
    try:
        line = main_hier(a, R)
    except Exception as e:   # noqa: BLE001 — reported, never silently dropped
        return {"error": f"{type(e).__name__}: {e}"}
    keys = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "gpu_ms_per_step", "per_rank_commander_steps_per_s", "sim_ticks_per_s", "ticks_per_commander_step",
            "ranks_seen", "gathered_rows", "first_global_arena_of_each_block", "arenas_of_each_block", "backend")
    out = {k: line[k] for k in keys if k in line}
    out["workload"] = line["config"]["workload"]
    out["parallelism"] = line["config"]["parallelism"]
    out["roofline"] = line["roofline"]
    return out


def extra_configs(args, R):
    """BASELINE configs[2] (16384 arenas, fight networks in the loop every tick) and configs[3] (8192 arenas x 3-vs-3 HighLevelEnv:
    pilot actions from a tape = one persistent launch per commander step, and with the reference's pilot networks in the loop),
    each measured by its own `--workload` run of this script in a CHILD process (shorter than a stand-alone run): whatever happens
    there — an exception, a crash — costs the headline nothing but an `error` entry."""
    def brief(line):
        keys = ("metric", "value", "unit", "steps", "ms_per_step", "gpu_ms_per_step", "dtype", "kernels_ms", "launches_per_step", "launches_per_sub_world_step", "sim_ticks_per_s",
                "ticks_per_commander_step", "streams", "collect", "agent_steps_per_s")
        out = {k: line[k] for k in keys if k in line}
        out["workload"] = line["config"]["workload"]
        for k in ("actions", "ticks_per_step", "timed_ticks_per_arena", "pilot_rows"):
            if k in line["config"]:
                out[k] = line["config"][k]
        out["roofline"] = line["roofline"]
        return out

    extra = {}
    R.torch.cuda.synchronize()
    t_extra, budget_s = time.perf_counter(), float(os.environ.get("HH_BENCH_EXTRA_BUDGET_S", "120"))   # the riders may not push the line past a few minutes
    for name, flags in (("configs1_saturated", ["--workload", "low", "--arenas", "262144", "--chunk", "125", "--steps", "8", "--warmup", "2", "--no-extra"]),
                        ("configs2", ["--workload", "rollout", "--ppo", "--steps", "300", "--warmup", "30"]),
                        ("configs2_collect", ["--workload", "collect", "--chunk", "64", "--steps", "6", "--warmup", "2"]),
                        ("configs2_greedy_inference", ["--workload", "rollout", "--steps", "300", "--warmup", "30"]),
                        ("configs3", ["--workload", "hier", "--pilot", "tape", "--steps", "40", "--warmup", "8"]),
                        ("configs3_networks_in_loop", ["--workload", "hier", "--pilot", "net", "--steps", "40", "--warmup", "5"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--spinup", "0.3", "--seed", str(args.seed), "--no-cpu-baseline"] + flags
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        left = budget_s - (time.perf_counter() - t_extra)
        if left < 10.0:
            extra[name] = {"skipped": f"the extra runs' time budget ({budget_s:.0f} s, HH_BENCH_EXTRA_BUDGET_S) was spent; run `python bench.py {' '.join(flags)}`"}
            continue
        t_child = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=min(90.0, left), env=env)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                extra[name] = {"error": f"child exited with {p.returncode}: {(p.stderr or '')[-300:]}"}
            else:
                extra[name] = brief(json.loads(lines[-1]))
                extra[name]["child_seconds"] = round(time.perf_counter() - t_child, 1)
        except Exception as e:   # noqa: BLE001 — reported, never silently dropped
            extra[name] = {"error": f"{type(e).__name__}: {e}"}
    return extra


if __name__ == "__main__":
    main()
