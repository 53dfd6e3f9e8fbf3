"""Batched tensor API of the MI355X air-combat world (thin host layer over include/hh_abi.h).

PyTorch is used for device memory and streams only; every step runs in the hand-written HIP
kernels of hhmarl_2d_amd/csrc/hh_world.hip."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def make_config(n_arenas=1, env_kind=L.ENV_LOWLEVEL, level=1, agent_mode=L.MODE_FIGHT, n_agents=None, n_opps=None,
                horizon=None, friendly_kill=True, friendly_punish=False, esc_dist_rew=False, hier_action_assess=True,
                hier_opp_fight_ratio=75, auto_reset=False, ext_opp_actions=False, map_size=None, glob_frac=0.0,
                rew_scale=1.0, seed=0, arena_offset=0, opp_side_selector=False):
    """Defaults follow the reference's config.py:17-54 and horizons config.py:94-98."""
    hl = env_kind == L.ENV_HIGHLEVEL
    if n_agents is None:
        n_agents = 3 if hl else 2
    if n_opps is None:
        n_opps = 3 if hl else 2
    if horizon is None:
        horizon = 500 if hl else {1: 150, 2: 200, 3: 300, 4: 350, 5: 400}[level]
    if map_size is None:
        map_size = 0.5 if hl else 0.3
    return L.HHConfig(n_arenas, env_kind, n_agents, n_opps, level, agent_mode, horizon, int(friendly_kill),
                      int(friendly_punish), int(esc_dist_rew), int(hier_action_assess), hier_opp_fight_ratio,
                      int(auto_reset), int(ext_opp_actions), int(opp_side_selector), 0, map_size, glob_frac, rew_scale, seed, arena_offset)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def action_tape_uniform(seed, arena_offset, step0, T, N, n_units=2, device=0, out=None):
    """the keyed synthetic action tape of the benchmark workloads (hh_action_tape_uniform: i.i.d. uniform over MultiDiscrete([13,9,2,2]),
    key = (seed, global arena, step, agent)) -> int8 [T, N, n_units, 4] on `device`"""
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    out = out if out is not None else torch.empty((T, N, n_units, 4), dtype=torch.int8, device=dev)
    assert out.is_contiguous() and out.dtype == torch.int8 and out.numel() == T * N * n_units * 4
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(L.lib().hh_action_tape_uniform(int(seed), int(arena_offset), int(step0), int(T), int(N), int(n_units), _p(out), st))
    return out


class World:
    """N independent arenas resident on one GPU."""

    def __init__(self, cfg, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("hhmarl_2d_amd.World needs a ROCm GPU (no CPU fallback)")
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        L.check(L.lib().hh_world_create(C.byref(cfg), device, C.byref(self.h)))
        self.N = cfg.n_arenas
        self.n_units = cfg.n_agents + cfg.n_opps
        # unit slots (HighLevelEnv: six, or ten with more than three aircraft on a side; unused ones never alive)
        self.A = L.hl_slots(cfg.n_agents, cfg.n_opps) if cfg.env_kind == L.ENV_HIGHLEVEL else self.n_units
        self.tgt_k = L.tgt_k_of(cfg.n_agents, cfg.n_opps) if cfg.env_kind == L.ENV_HIGHLEVEL else L.TGT_K
        self.n_agents = cfg.n_agents
        self.D = L.lib().hh_obs_dim(self.h)
        self.n_ctrl = L.lib().hh_n_ctrl(self.h)

    def close(self):
        if getattr(self, "h", None):
            L.lib().hh_world_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def alloc_outputs(self, T=None):
        lead = (self.N,) if T is None else (T, self.N)
        dev = self.device
        return (torch.zeros(lead + (self.n_agents, self.D), dtype=torch.float32, device=dev),
                torch.zeros(lead + (self.n_agents,), dtype=torch.float32, device=dev),
                torch.zeros(lead + (self.n_agents,), dtype=torch.uint8, device=dev),
                torch.zeros(lead, dtype=torch.uint8, device=dev))

    def reset(self, mask=None, obs=None):
        if obs is None:
            obs = torch.zeros((self.N, self.n_agents, self.D), dtype=torch.float32, device=self.device)
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        L.check(L.lib().hh_reset(self.h, _p(mask), _p(obs), self._stream()))
        return obs

    def observe(self, obs=None):
        if obs is None:
            obs = torch.zeros((self.N, self.n_agents, self.D), dtype=torch.float32, device=self.device)
        L.check(L.lib().hh_observe(self.h, _p(obs), self._stream()))
        return obs

    def step(self, actions, out=None):
        """actions: int8 [N, n_ctrl, 4] on the world's device -> (obs, reward, reward_valid, done)."""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.device == self.device
        assert actions.numel() == self.N * self.n_ctrl * 4
        obs, rew, val, done = out if out is not None else self.alloc_outputs()
        L.check(L.lib().hh_step(self.h, _p(actions), _p(obs), _p(rew), _p(val), _p(done), self._stream()))
        return obs, rew, val, done

    def step_begin(self, agent_actions, opp_mode=0, opp_obs=None):
        """levels 4-5: agents act; returns the frozen-policy opponents' observations f32 [N, n_opps, 30]"""
        assert agent_actions.dtype == torch.int8 and agent_actions.is_contiguous()
        assert agent_actions.numel() == self.N * self.n_agents * 4
        if opp_obs is None:
            opp_obs = torch.zeros((self.N, self.A - self.n_agents, 30), dtype=torch.float32, device=self.device)
        L.check(L.lib().hh_step_begin(self.h, _p(agent_actions), int(opp_mode), _p(opp_obs), self._stream()))
        return opp_obs

    def opp_policy(self, out=None):
        """level 5 / fight mode: k in {3,4,5} of every arena's current episode (env_hetero.py:55-59), int8 [N] on the device"""
        if out is None:
            out = torch.zeros((self.N,), dtype=torch.int8, device=self.device)
        L.check(L.lib().hh_opp_policy(self.h, _p(out), self._stream()))
        return out

    def step_finish(self, opp_actions, out=None):
        assert opp_actions.dtype == torch.int8 and opp_actions.is_contiguous()
        assert opp_actions.numel() == self.N * (self.A - self.n_agents) * 4
        obs, rew, val, done = out if out is not None else self.alloc_outputs()
        L.check(L.lib().hh_step_finish(self.h, _p(opp_actions), _p(obs), _p(rew), _p(val), _p(done), self._stream()))
        return obs, rew, val, done

    def rollout(self, actions, out=None, want_obs=True):
        """actions: int8 [T, N, n_ctrl, 4] pre-resident tape -> stacked outputs [T, ...]."""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.device == self.device
        T = actions.shape[0]
        assert actions.numel() == T * self.N * self.n_ctrl * 4
        obs, rew, val, done = out if out is not None else self.alloc_outputs(T)
        L.check(L.lib().hh_rollout(self.h, T, _p(actions), _p(obs) if want_obs else None, _p(rew), _p(val), _p(done),
                                   self._stream()))
        return obs, rew, val, done

    # ---- HighLevelEnv macro step (envs/env_hier.py:114-140); pilot inference runs between the calls
    def alloc_pilot(self):
        return (torch.zeros((self.N, self.A, 30), dtype=torch.float32, device=self.device),
                torch.zeros((self.N, self.A), dtype=torch.uint8, device=self.device))

    def hl_begin(self, commander_actions, pilot=None):
        """commander_actions int8 [N, n_agents] in {0,1,2} -> (pilot_obs [N,A,30], pilot_mode [N,A]) of the agents"""
        assert commander_actions.dtype == torch.int8 and commander_actions.is_contiguous()
        po, pm = pilot if pilot is not None else self.alloc_pilot()
        L.check(L.lib().hh_hl_begin(self.h, _p(commander_actions), _p(po), _p(pm), self._stream()))
        return po, pm

    def hl_agents_act(self, actions, pilot=None):
        """actions int8 [N, A, 4] (agent rows used) -> pilot observations of the opponents"""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.numel() == self.N * self.A * 4
        po, pm = pilot if pilot is not None else self.alloc_pilot()
        L.check(L.lib().hh_hl_agents_act(self.h, _p(actions), _p(po), _p(pm), self._stream()))
        return po, pm

    def hl_tick(self, actions, pilot=None, count_running=True):
        """actions int8 [N, A, 4] (opponent rows used) -> agents' pilot observations of the next sub-step and
        the number of arenas still inside their macro step (None when count_running=False: no host sync)"""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.numel() == self.N * self.A * 4
        po, pm = pilot if pilot is not None else self.alloc_pilot()
        running = C.c_int32(0)
        L.check(L.lib().hh_hl_tick(self.h, _p(actions), _p(po), _p(pm), C.byref(running) if count_running else None,
                                   self._stream()))
        return po, pm, (running.value if count_running else None)

    # ---- the variant-row form: one launch and one policy call per sub-step (include/hh_abi.h: hh_hl_begin_variants / hh_hl_act_tick)
    V_ROWS = 15   # row slots per arena: agents 0..2, opponent j's variant v at 3 + 4 j + v

    def alloc_pilot_variants(self):
        return (torch.zeros((self.N, self.V_ROWS, 30), dtype=torch.float32, device=self.device),
                torch.zeros((self.N, self.V_ROWS), dtype=torch.uint8, device=self.device))

    def hl_begin_variants(self, commander_actions, pilot=None):
        """commander_actions int8 [N, n_agents] -> (pilot_obs [N, 15, 30], pilot_mode [N, 15]): BOTH sides' rows of the first sub-step"""
        assert commander_actions.dtype == torch.int8 and commander_actions.is_contiguous()
        po, pm = pilot if pilot is not None else self.alloc_pilot_variants()
        L.check(L.lib().hh_hl_begin_variants(self.h, _p(commander_actions), _p(po), _p(pm), self._stream()))
        return po, pm

    def hl_act_tick(self, actions, pilot=None, count_running=True):
        """actions int8 [N, 15, 4] (the policy's output for every listed row) -> agents act, each opponent flies the variant that matches, tick;
        returns both sides' rows of the next sub-step and the number of arenas still inside their macro step"""
        assert actions.dtype == torch.int8 and actions.is_contiguous() and actions.numel() == self.N * self.V_ROWS * 4
        po, pm = pilot if pilot is not None else self.alloc_pilot_variants()
        running = C.c_int32(0)
        L.check(L.lib().hh_hl_act_tick(self.h, _p(actions), _p(po), _p(pm), C.byref(running) if count_running else None, self._stream()))
        return po, pm, (running.value if count_running else None)

    def hl_end(self, out=None):
        obs, rew, val, done = out if out is not None else self.alloc_outputs()
        L.check(L.lib().hh_hl_end(self.h, _p(obs), _p(rew), _p(val), _p(done), self._stream()))
        return obs, rew, val, done

    def hl_rollout(self, commander_actions, pilot_tape, out=None):
        """one whole commander step per arena in ONE launch: commander_actions int8 [N, n_agents], pilot_tape int8 [16, N, A, 4]"""
        assert commander_actions.dtype == torch.int8 and commander_actions.is_contiguous()
        assert pilot_tape.dtype == torch.int8 and pilot_tape.is_contiguous() and pilot_tape.numel() == 16 * self.N * self.A * 4
        obs, rew, val, done = out if out is not None else self.alloc_outputs()
        L.check(L.lib().hh_hl_rollout(self.h, _p(commander_actions), _p(pilot_tape), _p(obs), _p(rew), _p(val), _p(done), self._stream()))
        return obs, rew, val, done

    def hl_commands(self):
        """host int8 [N, A]: commander_actions after _action_assess (opponents' draws included)"""
        out = np.zeros((self.N, self.A), dtype=np.int8)
        L.check(L.lib().hh_hl_commands(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def episode_stats(self):
        ret = torch.zeros(self.N, dtype=torch.float32, device=self.device)
        ln = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        oc = torch.zeros(self.N, dtype=torch.int8, device=self.device)
        L.check(L.lib().hh_episode_stats(self.h, _p(ret), _p(ln), _p(oc), self._stream()))
        return ret, ln, oc

    def episode_stats_packed(self, out=None):
        """f32 [N, 3] (return, length, outcome) of the last finished episode per arena, one launch"""
        if out is None:
            out = torch.empty((self.N, 3), dtype=torch.float32, device=self.device)
        L.check(L.lib().hh_episode_stats_packed(self.h, _p(out), self._stream()))
        return out

    def eval_info(self, clear_total=False):
        """(last, total) int32 [N, 12] on the device: the reference's eval_info dict per arena for the most recent commander
        step and summed since the last clear; columns = _lib.EVAL_KEYS"""
        last = torch.empty((self.N, len(L.EVAL_KEYS)), dtype=torch.int32, device=self.device)
        tot = torch.empty_like(last)
        L.check(L.lib().hh_eval_info(self.h, _p(last), _p(tot), int(clear_total), self._stream()))
        return last, tot

    def action_faults(self, clear=False, out=None):
        """uint8 [N]: 1 where a step consumed an out-of-range action word since the last clear (it ran on the sanitised word:
        heading / speed component clamped, fire components as booleans) — the batched stand-in for the reference's raising guards"""
        out = out if out is not None else torch.zeros((self.N,), dtype=torch.uint8, device=self.device)
        L.check(L.lib().hh_action_faults(self.h, _p(out), int(bool(clear)), self._stream()))
        return out

    def arena_status(self):
        """int32 [N, 4] on the device: steps, alive_agents, alive_opps, done"""
        out = torch.empty((self.N, 4), dtype=torch.int32, device=self.device)
        L.check(L.lib().hh_arena_status(self.h, _p(out), self._stream()))
        return out

    def bind_policy(self, bank):
        """let the kernels that emit policy rows (HighLevelEnv phases; LowLevelEnv levels 4-5 step_begin) bin them into `bank`'s row
        lists (hh_bind_policy); None unbinds.  The world keeps the bank alive while bound."""
        L.check(L.lib().hh_bind_policy(self.h, bank.h if bank is not None else None))
        self._bound_bank = bank
        self.ptr_generation = getattr(self, "ptr_generation", 0) + 1   # captured HIP graphs hold the old pointers by value

    def trace_enable(self, n_arenas=1, capacity=1024):
        """device-side trajectory ring buffer for the first n_arenas arenas (0 turns it off)"""
        L.check(L.lib().hh_trace_enable(self.h, int(n_arenas), int(capacity)))
        self.ptr_generation = getattr(self, "ptr_generation", 0) + 1   # the old ring buffer was freed: captured graphs are stale
        self._trace_shape = (int(capacity), min(int(n_arenas), self.N), self.A, 8) if n_arenas and capacity else None

    def trace_read(self):
        """-> rows in time order per arena: list over traced arenas of float32 [T, A, 8] (lat, lon, hdg, spd, alive, rocket lat, rocket
        lon, rocket alive) and the episode number of every row (int [T])"""
        cap, K, A, F = self._trace_shape
        rows = np.zeros((cap, K, A, F), dtype=np.float32)
        cnt = np.zeros((K,), dtype=np.int32)
        L.check(L.lib().hh_trace_read(self.h, rows.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
        out = []
        for k in range(K):
            n = int(cnt[k])
            idx = [(i % cap) for i in range(max(0, n - cap), n)]
            r = rows[idx, k].copy()
            ep = (r[:, 0, 7] // 16).astype(np.int32)
            r[:, :, 7] = r[:, :, 7] % 16
            out.append((r, ep))
        return out

    def hl_tick_count(self):
        """cumulative arena-ticks run by macro steps on this world (synchronises the current stream)"""
        v = C.c_uint64(0)
        L.check(L.lib().hh_hl_tick_count(self.h, C.byref(v), self._stream()))
        return int(v.value)

    def kernel_name(self):
        buf = C.create_string_buffer(128)
        L.check(L.lib().hh_rollout_kernel_name(self.h, buf, 128))
        return buf.value.decode()

    def kernel_instance(self, which=0):
        """the launched instance as a profiler prints it (hh_kernel_instance); which = 1: the hh_hl_rollout kernel"""
        buf = C.create_string_buffer(128)
        L.check(L.lib().hh_kernel_instance(self.h, int(which), buf, 128))
        return buf.value.decode()

    # ---- host snapshots (parity tests, checkpointing) ----
    def _alloc_state(self):
        n, a = self.N, self.A
        return dict(
            ac_f=np.zeros((n, a, L.ACF_K)), ac_i=np.zeros((n, a, L.ACI_K), dtype=np.int32),
            rk_f=np.zeros((n, a, L.RKF_K)), rk_i=np.zeros((n, a, L.RKI_K), dtype=np.int32),
            ar_i=np.zeros((n, L.ARI_K), dtype=np.int32), tgt_id=np.zeros((n, a, self.tgt_k), dtype=np.int32),
            tgt_d=np.zeros((n, a, self.tgt_k)))

    @staticmethod
    def _view(st):
        def p(a, t):
            assert a.flags.c_contiguous
            return a.ctypes.data_as(C.POINTER(t))
        return L.HHStateView(p(st["ac_f"], C.c_double), p(st["ac_i"], C.c_int32), p(st["rk_f"], C.c_double),
                             p(st["rk_i"], C.c_int32), p(st["ar_i"], C.c_int32), p(st["tgt_id"], C.c_int32),
                             p(st["tgt_d"], C.c_double))

    def get_state(self):
        st = self._alloc_state()
        v = self._view(st)
        L.check(L.lib().hh_get_state(self.h, C.byref(v)))
        return st

    def set_state(self, st):
        st = {k: np.ascontiguousarray(v) for k, v in st.items()}
        v = self._view(st)
        L.check(L.lib().hh_set_state(self.h, C.byref(v)))

    def event_masks(self):
        m = np.zeros((self.N,), dtype=np.uint32)
        L.check(L.lib().hh_get_event_masks(self.h, m.ctypes.data_as(C.c_void_p), self._stream()))
        return m
