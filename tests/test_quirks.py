"""The load-bearing quirks of the reference, pinned one by one on hand-built
situations through the oracle's set_state (CPU).  The same situations run bit-for-bit on the GPU in
tests/test_gpu_parity.py::test_set_state_round_trip_and_edge_cases and through the golden traces."""
import numpy as np
import pytest


def _world(oracle, n=1, **kw):
    base = dict(n_arenas=n, level=3, seed=5, auto_reset=False)
    base.update(kw)
    w = oracle.OracleWorld(oracle.make_config(**base))
    w.reset()
    return w


def _place(st, n, slot, lat, lon, hdg, spd=100.0):
    st["ac_f"][n, slot] = (lat, lon, hdg, spd, hdg, spd)


def _noop(w, cannon=0, missile=0):
    a = np.zeros((w.N, w.n_ctrl, 4), dtype=np.int8)
    a[..., 0] = 6
    a[..., 2] = cannon
    a[..., 3] = missile
    return a


def test_q3_dead_this_tick_units_still_act_mutual_kill(oracle):
    """cmano_simulator.py:142: a unit killed earlier in the tick still shoots -> mutual cannon kill"""
    N = 400
    w = _world(oracle, N)
    st = w.get_state()
    for n in range(N):
        _place(st, n, 0, 5.15, 7.15, 90.0)
        _place(st, n, 2, 5.15, 7.16, 270.0)       # ~1.1 km apart, nose to nose
        _place(st, n, 1, 5.25, 7.05, 0.0)
        _place(st, n, 3, 5.05, 7.25, 180.0)
        st["ac_i"][n, 0, 3] = 5                     # bursts armed
        st["ac_i"][n, 2, 3] = 5
    w.set_state(st)
    w.step(_noop(w))
    m = w.event_masks()
    both = (m & 1).astype(bool) & ((m >> 2) & 1).astype(bool)
    only1 = ((m >> 2) & 1).astype(bool) & ~(m & 1).astype(bool)
    # P(hit) = 0.15 per shooter per tick, independent keyed draws: mutual kills must occur (p = 0.0225)
    assert both.sum() >= 2 and only1.sum() >= 20
    # agent 1 keeps its kill reward although it died in the same tick (reward key present, value > -2)
    obs, rew, val, done = w.step(_noop(w))
    assert (val[both, 0] == 0).all()               # dead at the start of the NEXT step: no key (Q12)


def test_q8_failed_launch_still_sets_missile_wait(oracle):
    """env_base.py:228-236: the gate passes, fire_missile fails the radar cone, wait is drawn and decremented"""
    w = _world(oracle, 64, level=1)
    st = w.get_state()
    for n in range(64):
        _place(st, n, 0, 5.15, 7.10, 270.0)        # target due east, nose west: outside (h-1, h+121)
        _place(st, n, 2, 5.15, 7.20, 0.0, 0.0)
        _place(st, n, 1, 5.28, 7.02, 0.0)
        _place(st, n, 3, 5.02, 7.28, 0.0, 0.0)      # farther than unit 3: the remembered target is unit 3
    w.set_state(st)
    assert (w.get_state()["ac_i"][:, 0, 9] == 3).all()
    w.step(_noop(w, missile=1))
    s2 = w.get_state()
    assert (s2["rk_i"][:, 0, 0] == 0).all() and (s2["ac_i"][:, 0, 5] == 5).all()   # no rocket, ammo intact
    wait = s2["ac_i"][:, 0, 7]
    assert wait.min() >= 6 and wait.max() <= 16 and len(set(wait.tolist())) > 3      # randint(7,17) - 1


def test_q6_asymmetric_radar_cone(oracle):
    """ac1.py:144-146: launch iff int(|sdiff(h+60, bearing)|) <= 60, i.e. bearing in (h-1, h+121)"""
    rel = np.array([-3.0, -1.5, -0.5, 0.0, 60.0, 120.0, 120.9, 121.5, 150.0])
    w = _world(oracle, len(rel), level=1)
    st = w.get_state()
    for n, r in enumerate(rel):
        _place(st, n, 0, 5.15, 7.15, 0.0)
        b = np.radians(r)
        _place(st, n, 2, 5.15 + 0.05 * np.cos(b), 7.15 + 0.05 * np.sin(b) / np.cos(np.radians(5.15)), 0.0, 0.0)
        _place(st, n, 1, 5.29, 7.01, 0.0)
        _place(st, n, 3, 5.01, 7.29, 0.0, 0.0)
    w.set_state(st)
    w.step(_noop(w, missile=1))
    launched = (w.event_masks() >> 24) & 1
    assert launched.tolist() == [0, 0, 1, 1, 1, 1, 1, 0, 0]


def test_q25_rocket_lives_11_flight_ticks_and_clears_one_tick_late(oracle):
    """rocket_unit.py:55-58 (.seconds > 10) and ac1.py:119-120 (actual_missile cleared one tick later)"""
    w = _world(oracle, 1, level=1, horizon=150)
    st = w.get_state()
    _place(st, 0, 0, 5.05, 7.05, 45.0)
    _place(st, 0, 2, 5.20, 7.20, 0.0, 0.0)          # 23 km away, in the cone: launch succeeds, never reached
    _place(st, 0, 1, 5.05, 7.28, 0.0)
    _place(st, 0, 3, 5.29, 7.29, 0.0, 0.0)
    w.set_state(st)
    alive, flag = [], []
    for t in range(15):
        w.step(_noop(w, missile=1 if t == 0 else 0))
        s = w.get_state()
        alive.append(int(s["rk_i"][0, 0, 0]))
        flag.append(int(s["ac_i"][0, 0, 8]))
    assert alive == [1] * 11 + [0] * 4             # 11 flight ticks, removed in the 12th update
    assert flag == [1] * 12 + [0] * 3              # launcher's flag clears one tick after the rocket is gone


def test_q28_out_of_bounds_is_inclusive_and_after_the_tick(oracle):
    """map_limits.py:47-48 / env_base.py:251-263: removed only when strictly outside, reward -5*s"""
    w = _world(oracle, 2, level=1, rew_scale=2.0)
    st = w.get_state()
    _place(st, 0, 0, 5.0005, 7.15, 180.0, 100.0)    # will cross lat 5.0 within a few ticks
    _place(st, 1, 0, 5.0005, 7.15, 0.0, 100.0)      # flies away from the border
    w.set_state(st)
    got = None
    for t in range(6):
        obs, rew, val, done = w.step(_noop(w))
        if (w.event_masks()[0] >> 16) & 1:
            got = (t, rew[0, 0])
            break
    assert got is not None and got[1] == -10.0
    assert w.get_state()["ac_i"][1, 0, 0] == 1


def test_q13_q12_observation_for_every_agent_and_reward_keys_only_for_alive(oracle):
    w = _world(oracle, 1, level=1)
    st = w.get_state()
    st["ac_i"][0, 1, 0] = 0                          # agent 2 already dead
    w.set_state(st)
    obs, rew, val, done = w.step(_noop(w))
    assert val[0].tolist() == [1, 0] and not obs[0, 1].any() and obs[0, 0].any()


def test_q21_commander_escape_targets_the_farthest_stored_enemy(oracle):
    """env_hier.py:130: commander action 0 indexes the stored list with -1"""
    w = oracle.OracleWorld(oracle.make_config(n_arenas=1, env_kind=oracle.ENV_HIGHLEVEL, seed=3))
    w.reset()
    st = w.get_state()
    assert st["tgt_id"][0, 0, 1] != 0               # agent 1 stores two enemies
    w.hl_begin(np.zeros((1, 3), dtype=np.int8))     # everybody escapes
    act = np.zeros((1, 6, 4), dtype=np.int8)
    act[..., 0] = 6
    act[0, 0, 3] = 1                                # agent 1 (type 1) pulls the missile trigger
    w.hl_agents_act(act)
    s2 = w.get_state()
    if s2["rk_i"][0, 0, 0]:                         # launch needs the radar cone; when it happens the target is the LAST entry
        assert s2["rk_i"][0, 0, 1] == st["tgt_id"][0, 0, 1]


def test_level3_escape_flag_is_consumed_once_per_opponent(oracle):
    """opps_escaping_time decrements once per live opponent call"""
    w = _world(oracle, 200, level=3, horizon=300)
    for t in range(60):
        w.step(_noop(w))
    s = w.get_state()["ar_i"]
    esc = s[:, 3] == 1
    assert esc.any() and (~esc).any()
    t0 = s[esc, 4].copy()
    alive_opps = w.get_state()["ac_i"][esc][:, 2:, 0].sum(axis=1)
    w.step(_noop(w))
    s1 = w.get_state()["ar_i"]
    still = (s1[esc, 3] == 1) & (s1[esc, 0] == s[esc, 0] + 1)   # still escaping, arena not finished
    assert still.sum() > 10
    assert np.array_equal((t0 - s1[esc, 4])[still], alive_opps[still])
