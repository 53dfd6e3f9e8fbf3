"""
TEST INFRASTRUCTURE — ctypes driver for oracle/libhh_oracle.so (the plain-C CPU restatement).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhh_oracle.so")

ENV_LOWLEVEL, ENV_HIGHLEVEL = 0, 1
MODE_FIGHT, MODE_ESCAPE = 0, 1


class HHConfig(C.Structure):
    """Mirror of hh_config in include/hh_abi.h (field order matters)."""
    _fields_ = [
        ("n_arenas", C.c_int32), ("env_kind", C.c_int32), ("n_agents", C.c_int32), ("n_opps", C.c_int32),
        ("level", C.c_int32), ("agent_mode", C.c_int32), ("horizon", C.c_int32), ("friendly_kill", C.c_int32),
        ("friendly_punish", C.c_int32), ("esc_dist_rew", C.c_int32), ("hier_action_assess", C.c_int32),
        ("hier_opp_fight_ratio", C.c_int32), ("auto_reset", C.c_int32), ("ext_opp_actions", C.c_int32),
        ("opp_side_selector", C.c_int32), ("reserved0", C.c_int32),
        ("map_size", C.c_double), ("glob_frac", C.c_double), ("rew_scale", C.c_double),
        ("seed", C.c_uint64), ("arena_offset", C.c_uint64),
    ]


class HHStateView(C.Structure):
    _fields_ = [
        ("ac_f", C.POINTER(C.c_double)), ("ac_i", C.POINTER(C.c_int32)), ("rk_f", C.POINTER(C.c_double)),
        ("rk_i", C.POINTER(C.c_int32)), ("ar_i", C.POINTER(C.c_int32)), ("tgt_id", C.POINTER(C.c_int32)),
        ("tgt_d", C.POINTER(C.c_double)),
    ]


ACF_K, ACI_K, RKF_K, RKI_K, ARI_K, TGT_K = 6, 10, 4, 4, 6, 3


def make_config(n_arenas=1, env_kind=ENV_LOWLEVEL, level=1, agent_mode=MODE_FIGHT, n_agents=None, n_opps=None,
                horizon=None, friendly_kill=True, friendly_punish=False, esc_dist_rew=False, hier_action_assess=True,
                hier_opp_fight_ratio=75, auto_reset=False, ext_opp_actions=False, map_size=None, glob_frac=0.0,
                rew_scale=1.0, seed=0, arena_offset=0, opp_side_selector=False):
    """Defaults follow config.py:17-54,94-98 of the reference."""
    hl = env_kind == ENV_HIGHLEVEL
    if n_agents is None:
        n_agents = 3 if hl else 2
    if n_opps is None:
        n_opps = 3 if hl else 2
    if horizon is None:
        horizon = 500 if hl else {1: 150, 2: 200, 3: 300, 4: 350, 5: 400}[level]
    if map_size is None:
        map_size = 0.5 if hl else 0.3
    return HHConfig(n_arenas, env_kind, n_agents, n_opps, level, agent_mode, horizon, int(friendly_kill),
                    int(friendly_punish), int(esc_dist_rew), int(hier_action_assess), hier_opp_fight_ratio,
                    int(auto_reset), int(ext_opp_actions), int(opp_side_selector), 0, map_size, glob_frac, rew_scale, seed, arena_offset)


def tgt_k_of(n_agents, n_opps):
    """entries of a stored target list in the state views (include/hh_spec.h: HH_TGT_K_OF)"""
    return 5 if max(n_agents, n_opps) > 3 else TGT_K


def alloc_state(n, a, k=TGT_K):
    return dict(
        ac_f=np.zeros((n, a, ACF_K)), ac_i=np.zeros((n, a, ACI_K), dtype=np.int32), rk_f=np.zeros((n, a, RKF_K)),
        rk_i=np.zeros((n, a, RKI_K), dtype=np.int32), ar_i=np.zeros((n, ARI_K), dtype=np.int32),
        tgt_id=np.zeros((n, a, k), dtype=np.int32), tgt_d=np.zeros((n, a, k)),
    )


def state_view(st):
    def p(a, t):
        assert a.flags.c_contiguous
        return a.ctypes.data_as(C.POINTER(t))
    return HHStateView(p(st["ac_f"], C.c_double), p(st["ac_i"], C.c_int32), p(st["rk_f"], C.c_double),
                       p(st["rk_i"], C.c_int32), p(st["ar_i"], C.c_int32), p(st["tgt_id"], C.c_int32),
                       p(st["tgt_d"], C.c_double))


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.hho_rng_u01.restype = C.c_double
        _lib.hho_rng_u01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.hho_omp_set_threads.restype = None
    return _lib


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class OracleWorld:
    """Batched CPU world with the same call surface as hhmarl_2d_amd.world.World (numpy in/out)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = lib().hho_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise ValueError(f"hho_create failed: {rc}")
        self.N = cfg.n_arenas
        self.A = cfg.n_agents + cfg.n_opps
        self.n_agents = cfg.n_agents
        self.D = lib().hho_obs_dim(self.h)
        self.n_ctrl = lib().hho_n_ctrl(self.h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            try:
                lib().hho_destroy(self.h)
            except Exception:
                pass
            self.h = None

    def reset(self, mask=None):
        obs = np.zeros((self.N, self.n_agents, self.D), dtype=np.float32)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        lib().hho_reset(self.h, _ptr(mask, C.c_uint8), _ptr(obs, C.c_float))
        if mask is not None:
            lib().hho_get_obs(self.h, _ptr(obs, C.c_float))
        return obs

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int8).reshape(self.N, self.n_ctrl, 4)
        obs = np.zeros((self.N, self.n_agents, self.D), dtype=np.float32)
        rew = np.zeros((self.N, self.n_agents), dtype=np.float32)
        val = np.zeros((self.N, self.n_agents), dtype=np.uint8)
        done = np.zeros((self.N,), dtype=np.uint8)
        rc = lib().hho_step(self.h, _ptr(actions, C.c_int8), _ptr(obs, C.c_float), _ptr(rew, C.c_float),
                            _ptr(val, C.c_uint8), _ptr(done, C.c_uint8))
        assert rc == 0, rc
        return obs, rew, val, done

    def step_begin(self, agent_actions, opp_mode=0):
        a = np.ascontiguousarray(agent_actions, dtype=np.int8).reshape(self.N, self.n_agents, 4)
        oo = np.zeros((self.N, self.A - self.n_agents, 30), dtype=np.float32)
        rc = lib().hho_step_begin(self.h, _ptr(a, C.c_int8), opp_mode, _ptr(oo, C.c_float))
        assert rc == 0, rc
        return oo

    def opp_policy(self):
        k = np.zeros((self.N,), dtype=np.int8)
        lib().hho_opp_policy(self.h, _ptr(k, C.c_int8))
        return k

    def step_finish(self, opp_actions):
        a = np.ascontiguousarray(opp_actions, dtype=np.int8).reshape(self.N, self.A - self.n_agents, 4)
        obs = np.zeros((self.N, self.n_agents, self.D), dtype=np.float32)
        rew = np.zeros((self.N, self.n_agents), dtype=np.float32)
        val = np.zeros((self.N, self.n_agents), dtype=np.uint8)
        done = np.zeros((self.N,), dtype=np.uint8)
        rc = lib().hho_step_finish(self.h, _ptr(a, C.c_int8), _ptr(obs, C.c_float), _ptr(rew, C.c_float),
                                   _ptr(val, C.c_uint8), _ptr(done, C.c_uint8))
        assert rc == 0, rc
        return obs, rew, val, done

    def alloc_rollout_outputs(self, T):
        """the four output arrays of rollout(), uninitialised (np.empty: pages are first touched by the threads that write them)"""
        return (np.empty((T, self.N, self.n_agents, self.D), dtype=np.float32), np.empty((T, self.N, self.n_agents), dtype=np.float32),
                np.empty((T, self.N, self.n_agents), dtype=np.uint8), np.empty((T, self.N), dtype=np.uint8))

    def rollout(self, actions, out=None):
        """T steps of every arena; out = alloc_rollout_outputs(T) re-uses the caller's arrays (every element is written by the call)"""
        T = actions.shape[0]
        actions = np.ascontiguousarray(actions, dtype=np.int8).reshape(T, self.N, self.n_ctrl, 4)
        obs, rew, val, done = out if out is not None else self.alloc_rollout_outputs(T)
        assert obs.shape == (T, self.N, self.n_agents, self.D) and obs.flags.c_contiguous and done.shape == (T, self.N)
        rc = lib().hho_rollout(self.h, T, _ptr(actions, C.c_int8), _ptr(obs, C.c_float), _ptr(rew, C.c_float),
                               _ptr(val, C.c_uint8), _ptr(done, C.c_uint8))
        assert rc == 0, rc
        return obs, rew, val, done

    def get_state(self):
        st = alloc_state(self.N, self.A, tgt_k_of(self.n_agents, self.A - self.n_agents) if self.cfg.env_kind == ENV_HIGHLEVEL else TGT_K)
        v = state_view(st)
        lib().hho_get_state(self.h, C.byref(v))
        return st

    def set_state(self, st):
        st = {k: np.ascontiguousarray(v) for k, v in st.items()}
        v = state_view(st)
        lib().hho_set_state(self.h, C.byref(v))

    def get_obs(self):
        obs = np.zeros((self.N, self.n_agents, self.D), dtype=np.float32)
        lib().hho_get_obs(self.h, _ptr(obs, C.c_float))
        return obs

    def event_masks(self):
        m = np.zeros((self.N,), dtype=np.uint32)
        lib().hho_get_event_masks(self.h, _ptr(m, C.c_uint32))
        return m

    def action_faults(self, clear=False):
        """u8 [N]: a consumed action word of the arena was out of range and ran sanitised (sticky until cleared)"""
        m = np.zeros((self.N,), dtype=np.uint8)
        lib().hho_action_faults(self.h, _ptr(m, C.c_uint8), int(bool(clear)))
        return m

    # ---- HighLevelEnv macro step (env_hier.py:114-140), split so that pilot inference runs between ticks
    def hl_begin(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.int8).reshape(self.N, self.n_agents)
        rc = lib().hho_hl_begin(self.h, _ptr(cmd, C.c_int8))
        assert rc == 0, rc

    def hl_pilot_obs(self, side):
        """side 0: agents (before anybody acted); side 1: opponents (after hl_agents_act)"""
        obs = np.zeros((self.N, self.A, 30), dtype=np.float32)
        mode = np.zeros((self.N, self.A), dtype=np.uint8)
        lib().hho_hl_pilot_obs(self.h, side, _ptr(obs, C.c_float), _ptr(mode, C.c_uint8))
        return obs, mode

    def hl_agents_act(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int8).reshape(self.N, self.A, 4)
        lib().hho_hl_agents_act(self.h, _ptr(actions, C.c_int8))

    def hl_tick(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int8).reshape(self.N, self.A, 4)
        return lib().hho_hl_tick(self.h, _ptr(actions, C.c_int8))

    def hl_end(self):
        obs = np.zeros((self.N, self.n_agents, self.D), dtype=np.float32)
        rew = np.zeros((self.N, self.n_agents), dtype=np.float32)
        val = np.zeros((self.N, self.n_agents), dtype=np.uint8)
        done = np.zeros((self.N,), dtype=np.uint8)
        lib().hho_hl_end(self.h, _ptr(obs, C.c_float), _ptr(rew, C.c_float), _ptr(val, C.c_uint8), _ptr(done, C.c_uint8))
        return obs, rew, val, done

    def eval_info(self):
        last = np.zeros((self.N, 12), dtype=np.int32)
        tot = np.zeros((self.N, 12), dtype=np.int32)
        lib().hho_eval_info(self.h, _ptr(last, C.c_int32), _ptr(tot, C.c_int32))
        return last, tot

    def hl_cmd(self):
        cmd = np.zeros((self.N, self.A), dtype=np.int32)
        sub = np.zeros((self.N,), dtype=np.int32)
        lib().hho_hl_get_cmd(self.h, _ptr(cmd, C.c_int32), _ptr(sub, C.c_int32))
        return cmd, sub

    def episode_stats(self):
        ret = np.zeros((self.N,), dtype=np.float32)
        ln = np.zeros((self.N,), dtype=np.int32)
        oc = np.zeros((self.N,), dtype=np.int8)
        lib().hho_episode_stats(self.h, _ptr(ret, C.c_float), _ptr(ln, C.c_int32), _ptr(oc, C.c_int8))
        return ret, ln, oc


def omp_max_threads():
    """threads an OpenMP team of the oracle library gets (omp_get_max_threads)"""
    return int(lib().hho_omp_max_threads())


def omp_team_size():
    """threads that really ran a parallel region just now (omp_get_num_threads inside one)"""
    return int(lib().hho_omp_team_size())


def omp_set_threads(n):
    lib().hho_omp_set_threads(int(n))


def action_tape_uniform(seed, arena_offset, step0, T, N, n_units=2):
    """hh_action_tape_uniform on the host: int8 [T, N, n_units, 4], the benchmark's keyed uniform MultiDiscrete([13,9,2,2]) actions"""
    out = np.zeros((T, N, n_units, 4), dtype=np.int8)
    L = lib()
    L.hho_action_tape_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.hho_action_tape_uniform(int(seed), int(arena_offset), int(step0), int(T), int(N), int(n_units), out.ctypes.data_as(C.c_void_p))
    return out


def math_eval(fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float64)
    o0 = np.empty_like(a)
    o1 = np.empty_like(a)
    D = C.c_double
    lib().hho_math_eval(fn, len(a), _ptr(a, D), _ptr(b, D), _ptr(o0, D), _ptr(o1, D))
    return o0, o1


def geo_direct(lat, lon, azi, s):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat, lon, azi, s)]
    o0 = np.empty_like(arrs[0])
    o1 = np.empty_like(arrs[0])
    D = C.c_double
    lib().hho_geo_direct(len(o0), *[_ptr(x, D) for x in arrs], _ptr(o0, D), _ptr(o1, D))
    return o0, o1


def geo_move(lat, lon, azi, s):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat, lon, azi, s)]
    o0 = np.empty_like(arrs[0])
    o1 = np.empty_like(arrs[0])
    D = C.c_double
    lib().hho_geo_move(len(o0), *[_ptr(x, D) for x in arrs], _ptr(o0, D), _ptr(o1, D))
    return o0, o1


def geo_inverse(lat1, lon1, lat2, lon2):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat1, lon1, lat2, lon2)]
    o0 = np.empty_like(arrs[0])
    o1 = np.empty_like(arrs[0])
    D = C.c_double
    lib().hho_geo_inverse(len(o0), *[_ptr(x, D) for x in arrs], _ptr(o0, D), _ptr(o1, D))
    return o0, o1


def missile_cone_planar(lat1, lon1, hdg, lat2, lon2):
    """planar stage of the launch predicate next to the exact one: (pre, exact, beta_planar, beta_geodesic)"""
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat1, lon1, hdg, lat2, lon2)]
    n = len(arrs[0])
    pre = np.empty(n, dtype=np.int32)
    exact = np.empty(n, dtype=np.int32)
    bp = np.empty(n)
    bg = np.empty(n)
    D = C.c_double
    lib().hho_missile_cone_planar(n, *[_ptr(x, D) for x in arrs], _ptr(pre, C.c_int32), _ptr(exact, C.c_int32), _ptr(bp, D), _ptr(bg, D))
    return pre, exact, bp, bg


def cannon_cone_planar(ac_type, lat1, lon1, hdg, lat2, lon2):
    """planar 'certainly outside the cannon cone' stage next to the exact predicate: (outside, exact)"""
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat1, lon1, hdg, lat2, lon2)]
    n = len(arrs[0])
    out = np.empty(n, dtype=np.int32)
    exact = np.empty(n, dtype=np.int32)
    lib().hho_cannon_cone_planar(n, int(ac_type), *[_ptr(x, C.c_double) for x in arrs], _ptr(out, C.c_int32), _ptr(exact, C.c_int32))
    return out, exact


def geo_inverse_estimate(lat1, lon1, lat2, lon2):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (lat1, lon1, lat2, lon2)]
    o0 = np.empty_like(arrs[0])
    o1 = np.empty_like(arrs[0])
    D = C.c_double
    lib().hho_geo_inverse_estimate(len(o0), *[_ptr(x, D) for x in arrs], _ptr(o0, D), _ptr(o1, D))
    return o0, o1
