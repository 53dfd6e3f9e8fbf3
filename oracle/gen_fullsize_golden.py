"""
TEST INFRASTRUCTURE — an independent look at BASELINE size (VERDICT r3, weak point 1 / next-round item 7).

The HIP-vs-oracle comparisons at 4096 / 16384 arenas share include/hh_math.h, hh_geodesic.h, hh_rng.h with the product, so a defect in
a shared header would be invisible to them at size.  This script records, from the REAL reference (imported unchanged behind
oracle/ref_harness.py, geodesic = the libm-based oracle/geodesic_ref.py — nothing of hh_math.h), the trajectories of SIXTEEN arenas of the
bench's own world — BASELINE configs[1]: 4096 arenas x 2-vs-2 fight level 3, seed 1234, auto-reset — namely the first, the last, both
sides of workgroup boundaries and a few in between, 300 ticks each through every episode end and reset on the way, fed by an action tape
any test can rebuild without this script: actions of arena g = numpy default_rng([1234, g]).integers(0, [13, 9, 2, 2], (300, 2, 4)).
tests/test_gpu_fullsize.py runs the full 4096-arena world on the MI355X with that tape in those arenas (other arenas: anything) and
compares THOSE arenas with this record: observations <= 1e-6, positions / headings / speeds <= 1e-9, reward keys / done / integer state
exact.  tests/test_fullsize_oracle.py replays the same record through the C oracle on the CPU.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_fullsize_golden.py [--check]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "fullsize_l3.npz")
SEED, N_WORLD, T, CHUNK = 1234, 4096, 300, 50
ARENAS = [0, 7, 8, 1023, 1024, 1717, 2047, 2048, 2900, 3071, 3072, 3555, 4087, 4088, 4094, 4095]


def tape_of(arena, ticks=T):
    """the committed definition of the action tape (tests rebuild it the same way)"""
    return np.random.default_rng([SEED, int(arena)]).integers(0, [13, 9, 2, 2], (ticks, 2, 4)).astype(np.int8)


def generate():
    args = H.make_args(level=3)          # config.py defaults of curriculum level 3, fight: horizon 300, map 0.3, friendly fire on
    out = dict(obs=[], reward=[], valid=[], done=[], ac_f=[], ac_i=[], ar_i=[])
    for g in ARENAS:
        env = H.RefEnv("low", args, seed=SEED, arena=g)
        acts = tape_of(g)
        obs = env.reset()
        rows = dict(obs=[], reward=[], valid=[], done=[], ac_f=[], ac_i=[], ar_i=[])
        for t in range(T):
            a = {1: [int(x) for x in acts[t, 0]], 2: [int(x) for x in acts[t, 1, :3]]}
            obs, rew, term, trunc, info = env.step(a)
            done = bool(term["__all__"])
            r, v = np.zeros(2), np.zeros(2, dtype=np.uint8)
            for i, x in rew.items():
                r[i - 1], v[i - 1] = x, 1
            if done:                      # auto_reset: the world re-samples the arena inside the step; the tick's observation row is the new episode's first
                obs = env.reset()
            rows["obs"].append(env.obs_array(obs, 26)); rows["reward"].append(r); rows["valid"].append(v); rows["done"].append(int(done))
            if (t + 1) % CHUNK == 0:      # world snapshot (hh_get_state layout) every CHUNK ticks
                st = env.state()
                rows["ac_f"].append(st["ac_f"]); rows["ac_i"].append(st["ac_i"]); rows["ar_i"].append(st["ar_i"])
        for k in out:
            out[k].append(np.asarray(rows[k]))
        print(f"arena {g}: {int(np.sum(rows['done']))} episodes ended, final steps counter {int(rows['ar_i'][-1][0])}")
    res = {k: np.stack(v) for k, v in out.items()}          # [16, T, ...] / [16, T / CHUNK, ...]
    res["obs"] = res["obs"].astype(np.float32)
    res["arenas"] = np.asarray(ARENAS, dtype=np.int32)
    res["meta"] = np.array(json.dumps(dict(seed=SEED, n_world=N_WORLD, ticks=T, chunk=CHUNK, level=3, args={k: v for k, v in vars(args).items()})))
    return res


if __name__ == "__main__":
    data = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in data if k not in old.files or not np.array_equal(old[k], data[k])]
        print("full-size fixture reproduces" if not bad else f"DIFFERENT: {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
