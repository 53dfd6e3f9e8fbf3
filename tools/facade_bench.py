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
