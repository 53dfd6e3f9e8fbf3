"""The RLlib-facing rates as a tracked record: env-steps/s of `LowLevelVectorEnv` — RLlib's BaseEnv protocol, one iteration =
poll + try_reset of every finished sub-environment + send_actions with a {env_id: {agent_id: action}} dict, host arrays in and out — at
N = 1 / 256 / 4096 / 16384 sub-environments on one world, and of `LowLevelEnv(num_envs=N).step(dict)` (batched dict protocol) beside it.
Writes gpurun_out/vector_env_rates.json (copied to profiles/r05_vector_env_rates.json).  usage: vector_env_rates.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hhmarl_2d_amd.config import make_args
from hhmarl_2d_amd.env_hetero import LowLevelEnv
from hhmarl_2d_amd.vector_env import LowLevelVectorEnv

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "vector_env_rates.json")
rec = {"what": "LowLevelVectorEnv (RLlib BaseEnv protocol over one MI355X world, level 3 fight): poll + try_reset(finished) + send_actions per iteration, "
               "host dicts in and out; env-steps/s = N / seconds per iteration", "host_cores": len(os.sched_getaffinity(0)), "rates": []}
rng = np.random.default_rng(0)
for N in (1, 256, 4096, 16384):
    venv = LowLevelVectorEnv({"args": make_args(0, level=3), "num_envs": N, "seed": 1})
    acts = [{1: np.stack([rng.integers(0, 13, N), rng.integers(0, 9, N), rng.integers(0, 2, N), rng.integers(0, 2, N)], axis=1),
             2: np.stack([rng.integers(0, 13, N), rng.integers(0, 9, N), rng.integers(0, 2, N)], axis=1)} for _ in range(4)]
    per_env = [[{1: a[1][e], 2: a[2][e]} for e in range(N)] for a in acts]

    def it(k):
        obs, rew, term, trunc, info, _ = venv.poll()
        for e in obs:
            if term[e]["__all__"]:
                venv.try_reset(e)
        venv.send_actions(dict(enumerate(per_env[k % 4])))
    for k in range(5):
        it(k)
    S = 400 if N <= 256 else (60 if N <= 4096 else 20)
    t0 = time.perf_counter()
    for k in range(S):
        it(k)
    dt = (time.perf_counter() - t0) / S
    venv.stop()
    env = LowLevelEnv({"args": make_args(0, level=3), "num_envs": N, "seed": 1})
    env.reset()
    for k in range(5):
        env.step(acts[k % 4])
    t0 = time.perf_counter()
    for k in range(S):
        env.step(acts[k % 4])
    dt2 = (time.perf_counter() - t0) / S
    rec["rates"].append({"num_envs": N, "vector_us_per_iteration": dt * 1e6, "vector_env_steps_per_s": N / dt, "vector_us_per_sub_env": dt / N * 1e6,
                         "batched_dict_step_us": dt2 * 1e6, "batched_dict_env_steps_per_s": N / dt2})
    print(rec["rates"][-1], flush=True)
# the same surface for the 3-vs-3 commander environment, the Fight / Esc pilot networks in the loop (one commander step of every sub-environment per send_actions)
from hhmarl_2d_amd.pilots import NetPilot, VariantNetPilot
from hhmarl_2d_amd.vector_env import HighLevelVectorEnv
for rows, key in (("variants", "hier_rates"), ("sides", "hier_rates_two_calls_per_sub_step")):
    rec[key] = []
    for N in (1, 256, 4096):
        venv = HighLevelVectorEnv({"args": make_args(1, level=5, horizon=500), "num_envs": N, "seed": 1, "pilot": lambda po, pm: None})
        if rows == "variants":   # what the facade builds itself from a policy_dir: one launch + one policy call per sub-step
            venv.b.pilot = VariantNetPilot(venv.b.world, seed=2)
            venv.b._pbuf = venv.b.world.alloc_pilot_variants()
        else:
            venv.b.pilot = NetPilot(venv.b.world, seed=2)
        per_env = [{e: {k: int(rng.integers(3)) for k in (1, 2, 3)} for e in range(N)} for _ in range(4)]

        def ith(k):
            obs, rew, term, trunc, info, _ = venv.poll()
            for e in obs:
                if term[e]["__all__"]:
                    venv.try_reset(e)
            venv.send_actions(per_env[k % 4])
        for k in range(4):
            ith(k)
        S = 100 if N <= 256 else 20
        t0 = time.perf_counter()
        for k in range(S):
            ith(k)
        dt = (time.perf_counter() - t0) / S
        rec[key].append({"num_envs": N, "ms_per_iteration": dt * 1e3, "commander_steps_per_s": N / dt})
        print(rows, rec[key][-1], flush=True)
        venv.stop()
v = [r["vector_env_steps_per_s"] for r in rec["rates"]]
rec["monotone_in_num_envs"] = all(b >= a for a, b in zip(v, v[1:]))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec))
