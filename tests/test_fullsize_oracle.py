"""The full-size fixture (tests/golden/fullsize_l3.npz: sixteen arenas of the bench's own 4096-arena world — first, last, workgroup
boundaries — 300 ticks each through episode ends and resets, recorded from the REAL reference with the libm-based geodesic; generator
oracle/gen_fullsize_golden.py) replayed through the C oracle on the CPU: one single-arena world per recorded arena at its global id.
The GPU side (the whole 4096-arena world, those arenas compared) is tests/test_gpu_fullsize.py."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_l3.npz")
OBS_TOL, FLOAT_TOL = 1e-6, 1e-9


def tape_of(seed, arena, ticks):
    """the action tape of the fixture: arena g acts by numpy default_rng([seed, g]) (oracle/gen_fullsize_golden.py: tape_of)"""
    return np.random.default_rng([int(seed), int(arena)]).integers(0, [13, 9, 2, 2], (ticks, 2, 4)).astype(np.int8)


def compare_arena(g, k, obs, rew, val, done, snaps, where):
    """obs [T, 2, 26], rew / val [T, 2], done [T], snaps = list of (ac_f [4, 6], ac_i [4, 10], ar_i [>= 5]) every `chunk` ticks, for recorded arena k"""
    assert np.array_equal(done.astype(np.uint8), g["done"][k]), f"{where}: done flags"
    assert np.array_equal(val, g["valid"][k]), f"{where}: reward keys"
    assert np.abs(rew - g["reward"][k]).max() <= 1e-6, f"{where}: rewards"
    assert np.abs(obs - g["obs"][k]).max() <= OBS_TOL, f"{where}: observations ({np.abs(obs - g['obs'][k]).max():.2e})"
    for c, (ac_f, ac_i, ar_i) in enumerate(snaps):
        assert np.array_equal(ac_i, g["ac_i"][k, c]), f"{where}: aircraft ints at snapshot {c}"
        assert np.array_equal(np.asarray(ar_i)[:5], g["ar_i"][k, c][:5]), f"{where}: arena ints at snapshot {c}"
        assert np.abs(ac_f - g["ac_f"][k, c]).max() <= FLOAT_TOL, f"{where}: aircraft floats at snapshot {c} ({np.abs(ac_f - g['ac_f'][k, c]).max():.2e})"


def test_oracle_reproduces_the_reference_at_the_benchs_arena_ids(oracle):
    g = np.load(GOLD)
    meta = json.loads(str(g["meta"]))
    T, chunk = meta["ticks"], meta["chunk"]
    assert meta["n_world"] == 4096 and len(g["arenas"]) == 16 and {0, 4095} <= set(int(a) for a in g["arenas"]) and int(g["done"].sum()) >= 30
    for k, a in enumerate(g["arenas"]):
        w = oracle.OracleWorld(oracle.make_config(n_arenas=1, level=meta["level"], seed=meta["seed"], arena_offset=int(a), auto_reset=True))
        w.reset()
        tape = tape_of(meta["seed"], a, T)
        obs, rew, val, done, snaps = [], [], [], [], []
        for c in range(T // chunk):
            o, r, v, d = w.rollout(tape[c * chunk:(c + 1) * chunk, None])
            obs.append(o[:, 0]); rew.append(r[:, 0]); val.append(v[:, 0]); done.append(d[:, 0])
            st = w.get_state()
            snaps.append((st["ac_f"][0], st["ac_i"][0], st["ar_i"][0]))
        compare_arena(g, k, np.concatenate(obs), np.concatenate(rew), np.concatenate(val), np.concatenate(done), snaps, f"arena {int(a)}")
