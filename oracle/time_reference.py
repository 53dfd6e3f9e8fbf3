"""
TEST INFRASTRUCTURE — times the REAL reference environment on one core of the build container.

BASELINE.md §3 wants, beside every GPU number, the cost of the reference's own Python path.  The
reference cannot travel to the GPU box, so its rate is measured here (imported unchanged from
/root/reference behind oracle/ref_harness.py: the same stand-ins the golden traces use — keyed
random tape, geodesic from oracle/geodesic_ref.py in place of the absent geographiclib) and
committed as profiles/reference_cpu_rate.json; bench.py quotes that file as a labelled second
baseline entry ("reference_python") next to the C port it times live.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference.py [seconds per configuration]
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "profiles", "reference_cpu_rate.json")


def time_low(level, budget):
    args = H.make_args(level=level)
    env = H.RefEnv("low", args, seed=1234, arena=0)
    rng = np.random.default_rng(level)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        env.reset()
        done = False
        while not done:
            act = {1: [int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2)), int(rng.integers(2))],
                   2: [int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2))]}
            _, _, term, _, _ = env.step(act)
            done = term["__all__"]
            steps += 1
    return steps / (time.perf_counter() - t0)


def time_high(budget):
    args = H.make_args(mode=1)
    env = H.RefEnv("high", args, seed=1234, arena=0)
    rng = np.random.default_rng(7)
    cls = type(env.env)
    ticks = [0]

    def policy_actions(self_, policy_type, agent_id, unit):  # the frozen pilot networks are not shipped: uniform actions
        self_.lowlevel_state(policy_type, agent_id, unit=unit)
        return {agent_id: np.array([int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2)), int(rng.integers(2))])}

    cls._policy_actions = policy_actions
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        env.reset()
        done = False
        while not done:
            s0 = env.env.steps
            _, _, term, _, _ = env.step({i: int(rng.integers(3)) for i in (1, 2, 3)})
            ticks[0] += env.env.steps - s0
            done = term["__all__"]
            steps += 1
    dt = time.perf_counter() - t0
    return steps / dt, ticks[0] / dt


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    rates = {f"lowlevel_2v2_L{lv}": time_low(lv, budget) for lv in (1, 2, 3)}
    hl_steps, hl_ticks = time_high(budget)
    cpu = "unknown"
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    rec = {
        "what": "the reference environment itself (IDSIA/hhmarl_2D envs/ + warsim/, imported unchanged), random actions, "
                "full episodes incl. reset, ONE core; geographiclib replaced by the pure-Python Karney series of "
                "oracle/geodesic_ref.py (same order of cost as pure-Python geographiclib 2.0)",
        "where": f"build container: {cpu}, CPython {platform.python_version()}, numpy {np.__version__}",
        "cores": 1, "seconds_per_config": budget, "unit": "env-steps/s",
        "env_steps_per_s": rates,
        "highlevel_3v3_commander_steps_per_s": hl_steps, "highlevel_3v3_ticks_per_s": hl_ticks,
        "script": "oracle/time_reference.py",
    }
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))
