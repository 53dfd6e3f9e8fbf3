"""
TEST INFRASTRUCTURE — tests/golden/gae_vectors.npz: reward / reward-key / done streams cut from the committed reference traces
(tests/golden/env_*.npz, i.e. what the REAL reference returned step by step, incl. the steps after an agent's death where it
still returns an observation but no reward key), seeded value predictions, and the advantages / value targets RLlib 2.4's
postprocessing gives for them (oracle/gae_ref.py) next to hh_gae's masked convention.

Run:  python oracle/gen_gae_golden.py [--check]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gae_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "gae_vectors.npz")
STREAMS = [   # trace, gamma, lambda (train_hetero.py:216 / train_hier.py:186)
    ("env_l3_fight_pursuit_share.npz", 0.99, 0.95),
    ("env_l2_fight_pursuit.npz", 0.99, 0.95),
    ("env_l3_escape_shaping.npz", 0.99, 0.95),
    ("env_hl_pursuit_pilots.npz", 0.99, 1.0),
]


def generate():
    out = {}
    for k, (name, gamma, lam) in enumerate(STREAMS):
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        step = g["kind"] == 1
        reward = g["reward"][step].astype(np.float32)[:, None, :]          # [T, 1, nA]: one arena
        valid = g["valid"][step].astype(np.uint8)[:, None, :]
        done = g["done"][step].astype(np.uint8)[:, None]
        T, _, nA = reward.shape
        rng = np.random.default_rng([20240917, k])
        value = rng.standard_normal((T + 1, 1, nA)).astype(np.float32)
        adv, ret = gae_ref.rllib_stream(reward, valid, value, done, gamma, lam)
        adv_m, ret_m = gae_ref.masked_stream(reward, valid, value, done, gamma, lam)
        tag = name[4:-4]
        out.update({f"{tag}/reward": reward, f"{tag}/valid": valid, f"{tag}/done": done, f"{tag}/value": value,
                    f"{tag}/gamma_lambda": np.array([gamma, lam]), f"{tag}/adv_rllib": adv, f"{tag}/ret_rllib": ret,
                    f"{tag}/adv_masked": adv_m, f"{tag}/ret_masked": ret_m})
        dead_rows = int(((valid == 0) & (np.cumsum(done[:, :, None], axis=0) - done[:, :, None] < done.sum())).sum())
        print(f"{tag}: T={T} agents={nA} episodes={int(done.sum())} rows without a reward key inside complete episodes={dead_rows}")
    return out


if __name__ == "__main__":
    data = generate()
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in data if k not in old.files or not np.array_equal(old[k], data[k])]
        print("gae fixtures reproduce" if not bad else f"DIFFERENT: {bad}")
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
