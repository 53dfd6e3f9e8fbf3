"""The RLlib-facing dict protocol end to end (host numpy actions in, host numpy observations / rewards / done out): the
PCIe-inclusive rate of `LowLevelEnv(num_envs=N).step(action_dict)`, against the device-resident C-ABI paths.  usage: facade_bench.py [arenas]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hhmarl_2d_amd.config import make_args
from hhmarl_2d_amd.env_hetero import LowLevelEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = LowLevelEnv({"args": make_args(0, level=3), "num_envs": N, "seed": 1})
env.reset()
rng = np.random.default_rng(0)
acts = [{1: np.stack([rng.integers(0, 13, N), rng.integers(0, 9, N), rng.integers(0, 2, N), rng.integers(0, 2, N)], axis=1),
         2: np.stack([rng.integers(0, 13, N), rng.integers(0, 9, N), rng.integers(0, 2, N)], axis=1)} for _ in range(16)]
for k in range(20):
    env.step(acts[k % 16])
S = 300
t0 = time.perf_counter()
for k in range(S):
    obs, rew, term, trunc, info = env.step(acts[k % 16])
    done = term["__all__"]
    if np.any(done):
        pass   # RLlib would reset the finished sub-environments here; the rate below is the step path alone
dt = (time.perf_counter() - t0) / S
print(f"LowLevelEnv(num_envs={N}).step(dict): {dt * 1e6:.0f} us per call -> {N / dt / 1e6:.2f} M env-steps/s (host arrays in and out)")

if len(sys.argv) > 2 and sys.argv[2] == "hier":   # HighLevelEnv.step(dict) with the Fight / Esc networks as pilots (synthetic weights)
    import torch
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    from hhmarl_2d_amd.pilots import NetPilot
    from hhmarl_2d_amd.world import World
    holder = {}

    class LazyPilot:            # the pilot needs the env's world: built on first use
        def __call__(self, po, pm):
            if "p" not in holder:
                holder["p"] = NetPilot(hl.world, seed=1, bind=False)
            return holder["p"](po, pm)
    hl = HighLevelEnv({"args": make_args(1), "num_envs": N, "seed": 1, "pilot": LazyPilot()})
    hl.pilot = holder["p"] = NetPilot(hl.world, seed=1)   # the library's own pilot: batches above 64 arenas replay a HIP graph
    hl.reset()
    cmds = [{i: rng.integers(0, 3, N) for i in (1, 2, 3)} for _ in range(8)]
    for k in range(5):
        hl.step(cmds[k % 8])
    S = 40
    t0 = time.perf_counter()
    for k in range(S):
        hl.step(cmds[k % 8])
    dt = (time.perf_counter() - t0) / S
    print(f"HighLevelEnv(num_envs={N}).step(dict), networks in the loop: {dt * 1e3:.2f} ms per call -> {N / dt:.3g} commander-steps/s")

if len(sys.argv) > 2 and sys.argv[2] == "vector":   # the BaseEnv surface RLlib's sampler speaks: poll / try_reset / send_actions, {env_id: {agent_id: ...}}
    from hhmarl_2d_amd.vector_env import LowLevelVectorEnv
    venv = LowLevelVectorEnv({"args": make_args(0, level=3), "num_envs": N, "seed": 1})
    per_env = [[{1: a[1][e], 2: a[2][e]} for e in range(N)] for a in acts[:4]]
    def it(k):
        obs, rew, term, trunc, info, _ = venv.poll()
        for e in obs:
            if term[e]["__all__"]:
                venv.try_reset(e)
        venv.send_actions(dict(enumerate(per_env[k % 4])))
    for k in range(5):
        it(k)
    S = 200 if N <= 256 else 30
    t0 = time.perf_counter()
    for k in range(S):
        it(k)
    dt = (time.perf_counter() - t0) / S
    print(f"LowLevelVectorEnv(num_envs={N}) poll + try_reset + send_actions: {dt * 1e6:.0f} us per iteration -> {N / dt / 1e6:.3f} M env-steps/s "
          f"(RLlib BaseEnv protocol: {2 * N} observation arrays and {N} reward dicts built per iteration)")

if len(sys.argv) > 2 and sys.argv[2] == "hvector":   # the same surface for the 3-vs-3 commander environment, the pilot networks in the loop
    from hhmarl_2d_amd.pilots import NetPilot
    from hhmarl_2d_amd.vector_env import HighLevelVectorEnv
    venv = HighLevelVectorEnv({"args": make_args(1, level=5, horizon=500), "num_envs": N, "seed": 1, "pilot": lambda po, pm: None})
    venv.b.pilot = NetPilot(venv.b.world, seed=2)
    rng = np.random.default_rng(0)
    per_env = [{e: {k: int(rng.integers(3)) for k in (1, 2, 3)} for e in range(N)} for _ in range(4)]
    def it(k):
        obs, rew, term, trunc, info, _ = venv.poll()
        for e in obs:
            if term[e]["__all__"]:
                venv.try_reset(e)
        venv.send_actions(per_env[k % 4])
    for k in range(4):
        it(k)
    S = 100 if N <= 256 else 20
    t0 = time.perf_counter()
    for k in range(S):
        it(k)
    dt = (time.perf_counter() - t0) / S
    print(f"HighLevelVectorEnv(num_envs={N}) poll + try_reset + send_actions: {dt * 1e3:.2f} ms per iteration -> {N / dt:.3g} commander-steps/s "
          f"(RLlib BaseEnv protocol; one commander step = up to 16 ticks with the pilot networks in between)")
