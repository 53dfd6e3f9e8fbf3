import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from hhmarl_2d_amd.config import make_args
from hhmarl_2d_amd.pilots import VariantNetPilot
from hhmarl_2d_amd.vector_env import HighLevelVectorEnv
rng = np.random.default_rng(0)
for N in (1, 16, 64):
    for gf in (65, 1):
        venv = HighLevelVectorEnv({"args": make_args(1, level=5, horizon=500), "num_envs": N, "seed": 1, "pilot": lambda po, pm: None})
        venv.b.pilot = VariantNetPilot(venv.b.world, seed=2); venv.b._pbuf = venv.b.world.alloc_pilot_variants(); venv.b._graph_from = gf
        per_env = [{e: {k: int(rng.integers(3)) for k in (1, 2, 3)} for e in range(N)} for _ in range(4)]
        def ith(k):
            obs, rew, term, trunc, info, _ = venv.poll()
            for e in obs:
                if term[e]["__all__"]:
                    venv.try_reset(e)
            venv.send_actions(per_env[k % 4])
        for k in range(6): ith(k)
        t0 = time.perf_counter()
        for k in range(100): ith(k)
        dt = (time.perf_counter() - t0) / 100
        print(f"N={N} graph_from={gf}: {dt*1e3:.3f} ms per iteration", flush=True)
        venv.stop()
