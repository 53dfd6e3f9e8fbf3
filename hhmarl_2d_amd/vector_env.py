"""An RLlib-consumable surface for MANY arenas: the `BaseEnv` protocol (ray/rllib/env/base_env.py, RLlib 2.4) over one batched world.

The reference hands RLlib ONE `LowLevelEnv` per rollout worker (train_hetero.py:212-215: `num_rollout_workers`, one env each) and RLlib
wraps it into a `BaseEnv` itself (`convert_to_base_env` -> `MultiAgentEnvToBaseEnv`), whose sampler loop is

    obs, rewards, terminateds, truncateds, infos, off_policy_actions = base_env.poll()      # {env_id: {agent_id: value}}
    for env_id with terminateds[env_id]["__all__"] or truncateds[env_id]["__all__"]:
        obs, infos = base_env.try_reset(env_id)                                             # {env_id: {agent_id: obs}}, {env_id: {}}
    base_env.send_actions({env_id: {agent_id: action}})

`LowLevelVectorEnv` IS such a BaseEnv, for N arenas resident on the GPU: sub-environment i is arena i of one `World(num_envs = N)`.
`send_actions` packs every sub-environment's action dict into one [N, 2, 4] int8 tensor and takes ONE `hh_step`; `poll` hands out what
that step produced, in the reference's own dict semantics per sub-environment (observations for every agent id — zeros when dead —,
rewards only for ids alive at step start, `terminateds == truncateds == {"__all__": done}`, infos {}); `try_reset(env_id)` returns the
episode's first observation.  Finished arenas are re-sampled by ONE masked `hh_reset` right after the step that finished them (the draw
is keyed by arena and episode number, so it does not matter when it is made) and `try_reset` serves the cached rows: no device call per
sub-environment.  An RLlib user swaps `env=LowLevelEnv` for a registered creator that returns this object
(`tune.register_env("hh_vector", lambda cfg: LowLevelVectorEnv(cfg))`; a BaseEnv instance is passed through by `convert_to_base_env`)
with `env_config["num_envs"] = N` — INTEGRATION.md.

Sub-environment i draws from the keyed RNG as arena `arena_offset + i`, exactly like a single-arena `LowLevelEnv` created with
`env_config["arena_offset"] = arena_offset + i` and the same seed: tests/test_gpu_vector_env.py drives 64 of each side by side.
Levels 4-5 (frozen opponent policies, env_base.py:312-398) take `env_config["policy_dir"]` or `["opponent_policy"]` like LowLevelEnv and step in two halves.
"""
import gc

import numpy as np

from . import _lib as L
from . import spaces
from .env_hetero import OBS_AC1, OBS_AC2, OBS_ESC_AC1, OBS_ESC_AC2, config_from_args

try:  # subclass RLlib's class when ray is importable (it is not in the build image)
    from ray.rllib.env.base_env import BaseEnv as _Base  # pragma: no cover
except Exception:  # noqa: BLE001
    class _Base:
        pass


class _GpuBackend:
    """the batched world behind the adapter: numpy in, numpy out, pinned host mirrors, one synchronisation per call"""

    def __init__(self, cfg, device, args=None, policy_dir=None, opponent_policy=None):
        import torch
        from .world import World
        self.torch = torch
        self.world = World(cfg, device=device)
        w = self.world
        # levels 4-5: the opponents fly frozen policies between the two halves of the step (env_hetero.py:160-172), exactly as LowLevelEnv does it
        self.opp = opponent_policy
        self.args = args
        self.l5_draw = args is not None and args.level == 5 and args.agent_mode == "fight"
        self.opp_mode, self.opp_k = "fight", None       # level 5: arrays over the arenas, refreshed after every reset (the world's keyed draw)
        if args is not None and args.level >= 4 and self.opp is None:   # _get_policies("LowLevel"), env_base.py:312-332
            from .pilots import OpponentNets, PolicyBank
            bank = PolicyBank.from_reference_dir(w.device, policy_dir, "LowLevel", args, max_rows=w.N * 2)
            self.opp = OpponentNets(w, bank=bank, bind=True, skip_first=False)
        self.N, self.n_agents, self.D = w.N, w.n_agents, w.D
        self._act = torch.zeros((w.N, w.n_ctrl, 4), dtype=torch.int8, device=w.device)
        self._act_pin = torch.zeros((w.N, w.n_ctrl, 4), dtype=torch.int8).pin_memory()
        self.act_host = self._act_pin.numpy()
        self._out = w.alloc_outputs()
        self._out_pin = [torch.zeros(t.shape, dtype=t.dtype).pin_memory() for t in self._out]
        self._mask = torch.zeros((w.N,), dtype=torch.uint8, device=w.device)
        self._mask_pin = torch.zeros((w.N,), dtype=torch.uint8).pin_memory()
        self.mask_host = self._mask_pin.numpy()
        self._robs = torch.zeros((w.N, w.n_agents, w.D), dtype=torch.float32, device=w.device)
        self._robs_pin = torch.zeros((w.N, w.n_agents, w.D), dtype=torch.float32).pin_memory()

    def reset(self, masked):
        """re-sample the arenas flagged in mask_host (all when masked is False) -> observations [N, n_agents, D] (rows of other arenas stale)"""
        if masked:
            self._mask.copy_(self._mask_pin, non_blocking=True)
        self.world.reset(mask=self._mask if masked else None, obs=self._robs)
        self._robs_pin.copy_(self._robs, non_blocking=True)
        self.torch.cuda.current_stream(self.world.device).synchronize()
        if self.l5_draw:   # env_hetero.py:55-59: k = randint(3, 5) per episode; the opponents observe in escape mode iff k == 5
            self.opp_k = self.world.opp_policy().cpu().numpy()
            self.opp_mode = np.where(self.opp_k == 5, "escape", "fight")
        return self._robs_pin.numpy()

    def step(self):
        """one hh_step with the actions in act_host -> (obs, reward, valid, done) host arrays"""
        self._act.copy_(self._act_pin, non_blocking=True)
        if self.opp is not None:   # agents act, the frozen-policy opponents observe (the agents' same-tick weapon flags included) and act, then the tick
            w = self.world
            opp_obs = w.step_begin(self._act[:, : w.n_agents].contiguous(), L.OPP_MODE_EPISODE if self.l5_draw else 0)
            opp_act = self.opp(opp_obs, self).to(self.torch.int8).contiguous()
            w.step_finish(opp_act, out=self._out)
        else:
            self.world.step(self._act, out=self._out)
        for src, dst in zip(self._out, self._out_pin):
            dst.copy_(src, non_blocking=True)
        self.torch.cuda.current_stream(self.world.device).synchronize()
        return [t.numpy() for t in self._out_pin]

    def close(self):
        self.world.close()


class _NoCyclicGC:
    """The bulk builds below create tens of thousands of small dicts per call, none of them part of a cycle.  CPython's generational collector
    counts container allocations and runs a pass every 700 of them, promoting the survivors — with N = 4096 sub-environments that is what made
    the per-sub-environment cost GROW with N (2.5 us against 1.3 us at N = 256, same code).  Collection is paused for the duration of a build and
    the caller's setting restored afterwards (reference counting frees everything here anyway)."""

    def __enter__(self):
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            gc.enable()
        return False


class _VectorProtocol(_Base):
    """the BaseEnv call protocol over a backend with reset(masked) / step() / act_host / mask_host; subclasses name the agents, cut the observation rows
    and pack the actions (`_obs_dicts`, `_pack_all`, `_put_action`, `_info_of`).

    What this surface costs is Python objects — RLlib's protocol wants {env_id: {agent_id: value}} dicts for observations, rewards, terminateds,
    truncateds and infos of every sub-environment on every step — so everything per sub-environment is built in bulk: the action dicts are packed with
    one concatenate per agent id, the observation rows of an agent are cut with one slice and iterated once, rewards / done flags cross into Python as
    lists, and the five result dicts are kept as they will be polled (poll hands them over, it does not rebuild them).  Measured per iteration
    (poll + try_reset of the finished + send_actions) in profiles/r05_vector_env_rates.json."""

    def _init_protocol(self, backend, num_envs, agent_ids):
        self.b = backend
        self.num_envs = num_envs
        self._agent_ids = set(agent_ids)
        self._ids = sorted(self._agent_ids)
        self._all = dict.fromkeys(range(num_envs)).keys()   # every env id: `action_dict.keys() == self._all` is one C-speed comparison
        self._started = False
        self._new_pending()
        self._done = set()                    # sub-environments whose episode ended and that were not reset yet
        self._reset_obs = {}                  # env_id -> the first observation of its next episode (arena already re-sampled)
        self._fresh = set()                   # sub-environments that were reset and have not stepped since

    def _new_pending(self):
        # results nobody polled yet, as the five dicts poll returns: env_id -> obs dict | reward dict | terminateds | truncateds | infos
        self._p_obs, self._p_rew, self._p_term, self._p_trunc, self._p_info = {}, {}, {}, {}, {}

    def _info_of(self, e):
        return {}

    def _put_action(self, a, e, k, v):
        a[e, k - 1, : len(v)] = v

    def _pack_all(self, a, dicts):
        """actions of ALL sub-environments (dicts in env order) into the host mirror `a`; False = not applicable (odd shapes): per-row path"""
        return False

    def _obs_dicts(self, obs, envs):
        """observation dicts of the sub-environments `envs` (a range or a list) from the step's rows obs [N, n_agents, D] (a fresh copy: the
        per-agent arrays are views of it)"""
        raise NotImplementedError

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def action_space(self):
        return self._action_space

    def get_agent_ids(self):
        return self._agent_ids

    def get_sub_environments(self, as_dict=False):
        return {} if as_dict else []          # the arenas live in one device-resident world: there are no per-env Python objects

    def _start(self):
        with _NoCyclicGC():
            self._start_inner()

    def _start_inner(self):
        self._started = True
        rows = self.b.reset(False).copy()   # ONE copy out of the backend's reused host mirror; the per-agent observations are views of it
        envs = range(self.num_envs)
        od = self._obs_dicts(rows, envs)
        self._p_obs = dict(zip(envs, od))
        self._p_rew = {e: {} for e in envs}
        self._p_term = {e: {"__all__": False} for e in envs}
        self._p_trunc = dict(self._p_term)   # terminateds is truncateds, per sub-environment (env_base.py:108)
        self._p_info = {e: {} for e in envs}
        self._fresh = set(envs)

    def poll(self):
        """-> (obs, rewards, terminateds, truncateds, infos, off_policy_actions), each {env_id: {agent_id | "__all__": value}}, for every
        sub-environment with a result nobody polled yet (the first call resets them all)"""
        if not self._started:
            self._start()
        out = (self._p_obs, self._p_rew, self._p_term, self._p_trunc, self._p_info, {})
        self._new_pending()
        return out

    def send_actions(self, action_dict):
        """{env_id: {agent_id: action}}: every sub-environment whose episode is running must be there (the arenas of one world step
        together); one that ended needs try_reset first (RLlib's MultiAgentEnvToBaseEnv raises the same ValueError).

        A sub-environment that ended and was NOT try_reset before this call (RLlib's sampler always resets at once) has no action here: its
        arena — already re-sampled when its episode ended — moves along with the rest on an all-zero action, and is re-sampled once more
        after the step, so that the episode try_reset eventually hands out is an untouched one.  That costs the arena one episode number
        of the keyed RNG per skipped step: the equivalence with a single-arena LowLevelEnv at arena_offset + i holds for the call order
        poll -> try_reset(finished) -> send_actions, the one RLlib uses."""
        with _NoCyclicGC():
            self._send_actions(action_dict)

    def _send_actions(self, action_dict):
        n = self.num_envs
        full = not self._done and len(action_dict) == n and action_dict.keys() == self._all
        if not full:
            for e in action_dict:
                if e in self._done:
                    raise ValueError(f"Env {e} is already done and cannot accept new actions")
            missing = [e for e in range(n) if e not in action_dict and e not in self._done]
            if missing:
                raise ValueError(f"send_actions: sub-environments {missing[:8]} have a running episode but no action (the arenas of one world step together)")
        a = self.b.act_host
        packed = False
        if full:
            try:
                packed = self._pack_all(a, [action_dict[e] for e in range(n)])
            except (ValueError, TypeError, KeyError, IndexError):
                packed = False
        if not packed:
            a[:] = 0
            for e, ad in action_dict.items():
                for k, v in ad.items():
                    self._put_action(a, e, k, v)
        obs, rew, val, done = self.b.step()
        obs = obs.copy()                           # one copy of the step's observations: the agents' arrays are views of it
        envs = range(n) if full else list(action_dict)
        sel = slice(None) if full else envs
        ids = self._ids
        rew_l, done_l = rew[sel].tolist(), done[sel].tolist()
        if len(ids) == 2:
            i0, i1 = ids
            rd = [{i0: r[0], i1: r[1]} for r in rew_l]
        elif len(ids) == 3:
            i0, i1, i2 = ids
            rd = [{i0: r[0], i1: r[1], i2: r[2]} for r in rew_l]
        else:
            rd = [dict(zip(ids, r)) for r in rew_l]
        vsel = val[sel][:, : len(ids)]
        if not vsel.all():                          # rewards only for ids alive at step start: rebuild the few rows that lost a key
            for k in np.nonzero(~vsel.all(axis=1))[0].tolist():
                rd[k] = {i: rew_l[k][i - 1] for i in ids if vsel[k, i - 1]}
        dd = [{"__all__": bool(d)} for d in done_l]
        od = self._obs_dicts(obs, envs)
        self._fresh = set()
        if self._p_obs:                             # results of earlier steps nobody polled: the newer ones replace them per sub-environment
            self._p_obs.update(zip(envs, od)); self._p_rew.update(zip(envs, rd)); self._p_term.update(zip(envs, dd)); self._p_trunc.update(zip(envs, dd))
            self._p_info.update((e, self._info_of(e)) for e in envs)
        else:
            self._p_obs, self._p_rew, self._p_term = dict(zip(envs, od)), dict(zip(envs, rd)), dict(zip(envs, dd))
            self._p_trunc = dict(self._p_term)
            self._p_info = {e: self._info_of(e) for e in envs}
        fin = np.nonzero(done)[0].tolist() if full else [e for e, d in zip(envs, done_l) if d]
        self._done.update(fin)
        # see the docstring: sub-environments that ended earlier and were never try_reset were stepped along on zero actions
        fin += [e for e in self._done if e not in action_dict and e not in fin]
        if fin:   # ONE masked reset for every arena that just finished; try_reset hands out the cached rows
            m = self.b.mask_host
            m[:] = 0
            m[fin] = 1
            rows = self.b.reset(True).copy()
            self._reset_obs.update(zip(fin, self._obs_dicts(rows, fin)))

    def try_reset(self, env_id=None, *, seed=None, options=None):
        """-> ({env_id: obs dict}, {env_id: {}}) of the sub-environment's next episode.  RLlib's MultiAgentEnvToBaseEnv.try_reset calls
        env.reset(), i.e. EVERY try_reset starts a new episode; so does this one, with two cases in which the new episode already exists and no
        device call is made: the sub-environment's episode just ended (its arena was re-sampled by the step's one masked reset), or it was reset
        and that first observation has not even been polled yet.  Anything else — a reset in the middle of an episode, a second reset without a
        step in between — re-samples the arena (one masked hh_reset), exactly as calling reset() again on a single-arena LowLevelEnv would."""
        if env_id is None:
            env_id = 0
        if not self._started:
            self._start()
        if env_id in self._reset_obs:            # its episode ended: the arena was re-sampled right after that step
            o = self._reset_obs.pop(env_id)
        elif env_id in self._fresh and env_id in self._p_obs:   # reset, not stepped, not polled: that observation is the answer
            o = self._p_obs[env_id]
        else:                                    # a reset in the middle of an episode (RLlib does this for episodes it truncates itself)
            m = self.b.mask_host
            m[:] = 0
            m[env_id] = 1
            o = self._obs_dicts(self.b.reset(True).copy(), [env_id])[0]
        self._done.discard(env_id)
        for p in (self._p_obs, self._p_rew, self._p_term, self._p_trunc, self._p_info):
            p.pop(env_id, None)
        self._fresh.add(env_id)
        return {env_id: o}, {env_id: {}}

    def try_restart(self, env_id=None):
        return None

    def stop(self):
        self.b.close()

    close = stop


def _require_args(env_config, what):
    args = env_config.get("args", None)
    if args is None:
        raise ValueError(f"{what}: env_config['args'] is missing — pass the reference's config.Config(mode).get_arguments namespace (or "
                         "hhmarl_2d_amd.config.make_args(mode, level=..., ...)), exactly what train_hetero.py / train_hier.py put there")
    return args


class LowLevelVectorEnv(_VectorProtocol):
    """N LowLevelEnv sub-environments (2-vs-2) behind RLlib's BaseEnv protocol, one MI355X world underneath."""

    def __init__(self, env_config):
        self.args = _require_args(env_config, "LowLevelVectorEnv")
        if self.args.level >= 4 and env_config.get("opponent_policy") is None and env_config.get("policy_dir") is None and env_config.get("_backend") is None:
            raise ValueError("levels 4-5 fly frozen opponent policies (envs/env_base.py:312-398); pass env_config['policy_dir'] = the directory of the "
                             "exported L*_AC*_{fight,escape}.pt files, or env_config['opponent_policy'] = callable(opp_obs f32 [N,2,30] on the device, "
                             "backend) -> int8 actions [N,2,4] for units 3,4 of every sub-environment (backend.world, .opp_k, .opp_mode as in LowLevelEnv)")
        self.agent_mode = self.args.agent_mode
        fight = self.agent_mode == "fight"
        self.obs_dim_map = {1: OBS_AC1 if fight else OBS_ESC_AC1, 2: OBS_AC2 if fight else OBS_ESC_AC2}
        self._agent_ids = set(range(1, self.args.num_agents + 1))
        self.num_envs = int(env_config.get("num_envs", 1))
        # the spaces of ONE sub-environment, as the reference declares them (env_hetero.py:29-43)
        self._observation_space = spaces.Dict({i: spaces.Box(low=np.zeros(d), high=np.ones(d), dtype=np.float32) for i, d in self.obs_dim_map.items()})
        self._action_space = spaces.Dict({1: spaces.MultiDiscrete([13, 9, 2, 2]), 2: spaces.MultiDiscrete([13, 9, 2])})
        backend = env_config.get("_backend", None)   # tests: a CPU stand-in with the same reset / step surface
        if backend is None:
            cfg = config_from_args(self.args, L.ENV_LOWLEVEL, self.num_envs, int(env_config.get("seed", 0)), auto_reset=False,
                                   arena_offset=int(env_config.get("arena_offset", 0)))
            backend = _GpuBackend(cfg, int(env_config.get("device", 0)), self.args, env_config.get("policy_dir"), env_config.get("opponent_policy"))
        self._init_protocol(backend, self.num_envs, range(1, self.args.num_agents + 1))

    def _obs_dicts(self, obs, envs):
        d1, d2 = self.obs_dim_map[1], self.obs_dim_map[2]
        if isinstance(envs, range):   # all of them: one slice per agent id, iterated once (row views of the caller's fresh copy)
            return [{1: x, 2: y} for x, y in zip(obs[:, 0, :d1], obs[:, 1, :d2])]
        return [{1: obs[e, 0, :d1], 2: obs[e, 1, :d2]} for e in envs]

    def _pack_all(self, a, dicts):
        n = len(dicts)
        a1 = np.concatenate([d[1] for d in dicts])   # agent 1: MultiDiscrete([13, 9, 2, 2]); agent 2 (type 2, no missile): ([13, 9, 2])
        a2 = np.concatenate([d[2] for d in dicts])
        if a1.size != 4 * n or a2.size != 3 * n or not all(len(d) == 2 for d in dicts):   # a dict with other agent ids: the per-row path decides what it means
            return False
        a[:, 0, :] = a1.reshape(n, 4)
        a[:, 1, :3] = a2.reshape(n, 3)
        a[:, 1, 3] = 0
        return True


class _GpuHierBackend:
    """the batched 3-vs-3 HighLevelEnv world behind HighLevelVectorEnv: one commander step of every sub-environment per step() — the 34 launches of
    env_hier.macro_step (variant rows; 66 with pilot_rows = "sides"), replayed from ONE HIP graph when the library's own NetPilot flies the units, eager with the early exit for a handful of arenas"""

    def __init__(self, cfg, device, args, policy_dir=None, pilot=None, graph_from=1, pilot_rows="variants"):
        import torch
        from .env_hier import macro_step
        from .world import World
        self.torch, self._macro_step = torch, macro_step
        self.world = World(cfg, device=device)
        w = self.world
        self.N, self.n_agents, self.D = w.N, w.n_agents, w.D
        self.pilot = pilot
        if self.pilot is None:   # _get_policies("HighLevel"), env_base.py:333-343
            from .pilots import own_pilot
            self.pilot = own_pilot(w, policy_dir, args, pilot_rows)
        self._cmd = torch.zeros((w.N, w.n_agents), dtype=torch.int8, device=w.device)
        self._cmd_pin = torch.zeros((w.N, w.n_agents), dtype=torch.int8).pin_memory()
        self.act_host = self._cmd_pin.numpy()
        self._out = w.alloc_outputs()
        self._pbuf = w.alloc_pilot_variants() if getattr(self.pilot, "variants", False) else w.alloc_pilot()
        self._out_pin = [torch.zeros(t.shape, dtype=t.dtype).pin_memory() for t in self._out]
        self._mask = torch.zeros((w.N,), dtype=torch.uint8, device=w.device)
        self._mask_pin = torch.zeros((w.N,), dtype=torch.uint8).pin_memory()
        self.mask_host = self._mask_pin.numpy()
        self._robs = torch.zeros((w.N, w.n_agents, w.D), dtype=torch.float32, device=w.device)
        self._robs_pin = torch.zeros((w.N, w.n_agents, w.D), dtype=torch.float32).pin_memory()
        self._graph, self._graph_gen, self._graph_from = None, -1, graph_from

    def reset(self, masked):
        if masked:
            self._mask.copy_(self._mask_pin, non_blocking=True)
        self.world.reset(mask=self._mask if masked else None, obs=self._robs)
        self._robs_pin.copy_(self._robs, non_blocking=True)
        self.torch.cuda.current_stream(self.world.device).synchronize()
        return self._robs_pin.numpy()

    def _replay(self):
        from .pilots import NetPilot, VariantNetPilot
        torch, w = self.torch, self.world
        if not isinstance(self.pilot, (NetPilot, VariantNetPilot)):   # a foreign pilot may read things on the host: no capture
            return self._macro_step(w, self._cmd, self.pilot, out=self._out, pilot_buf=self._pbuf, early_exit=False)
        # the graph holds device pointers by value: the world's (trace ring, bound bank's row lists) and the bank's (weight blobs, selector table)
        gen = (getattr(w, "ptr_generation", 0), id(self.pilot.bank), getattr(self.pilot.bank, "generation", 0))
        if self._graph is None or self._graph_gen != gen:
            nw = min(64, self.pilot.bank.max_rows)
            self.pilot.bank.act(torch.zeros((nw, 30), device=w.device), torch.zeros((nw,), dtype=torch.uint8, device=w.device))   # first launches outside a capture
            torch.cuda.synchronize(w.device)
            side = torch.cuda.Stream(device=w.device)
            side.wait_stream(torch.cuda.current_stream(w.device))
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    self._macro_step(w, self._cmd, self.pilot, out=self._out, pilot_buf=self._pbuf, early_exit=False)
            torch.cuda.current_stream(w.device).wait_stream(side)
            self._graph, self._graph_gen = graph, gen
        self._graph.replay()
        return self._out

    def step(self):
        """one commander step (HighLevelEnv.step, env_hier.py:114-140) with the actions in act_host -> (obs, reward, valid, done) host arrays"""
        self._cmd.copy_(self._cmd_pin, non_blocking=True)
        from .pilots import NetPilot, VariantNetPilot
        own = isinstance(self.pilot, (NetPilot, VariantNetPilot))
        if not own or self.N < self._graph_from:   # a foreign pilot (or graph_from raised): eager launches, leaving the sub-step loop early for a handful of arenas
            outs = self._macro_step(self.world, self._cmd, self.pilot, out=self._out, pilot_buf=self._pbuf, early_exit=self.N < 65)
        else:   # the library's own pilot: one replayed HIP graph at every size (0.65 against 1.03 ms for a single sub-environment: tools/hl_small_batch_probe.py)
            outs = self._replay()
        for src, dst in zip(outs, self._out_pin):
            dst.copy_(src, non_blocking=True)
        self.torch.cuda.current_stream(self.world.device).synchronize()
        return [t.numpy() for t in self._out_pin]

    def eval_info(self):
        return self.world.eval_info()[0].cpu().numpy()

    def close(self):
        self.world.close()


class HighLevelVectorEnv(_VectorProtocol):
    """N HighLevelEnv sub-environments (3-vs-3 commander, envs/env_hier.py) behind RLlib's BaseEnv protocol, one MI355X world underneath: what
    train_hier.py's `env=HighLevelEnv` (one env per rollout worker, train_hier.py:196-199) becomes with thousands of arenas per GPU.  One `send_actions`
    = one commander step of every sub-environment (up to 16 ticks with the frozen pilot networks in between).  env_config as HighLevelEnv's
    (`policy_dir` or `pilot`), plus `num_envs`, `seed`, `arena_offset`, `device`; `args.eval_info` puts the reference's win / lose / draw and choice counters
    (env_base.py:91-107) into every sub-environment's info dict."""

    def __init__(self, env_config):
        from .env_hier import N_OPP_HL, OBS_HL
        self.args = _require_args(env_config, "HighLevelVectorEnv")
        if env_config.get("pilot") is None and env_config.get("policy_dir") is None and env_config.get("_backend") is None:
            raise ValueError("HighLevelEnv flies frozen low-level pilot policies (envs/env_base.py:312-398); pass env_config['policy_dir'] = the directory of the "
                             "exported L*_AC*_{fight,escape}.pt files, or env_config['pilot'] = callable(pilot_obs, pilot_mode) -> int8 actions [N, A, 4]")
        self.num_envs = int(env_config.get("num_envs", 1))
        self._observation_space = spaces.Box(low=np.zeros(OBS_HL), high=np.ones(OBS_HL), dtype=np.float32)   # per agent, as the reference declares it (env_hier.py:34-35)
        self._action_space = spaces.Discrete(N_OPP_HL + 1)
        backend = env_config.get("_backend", None)
        if backend is None:
            cfg = config_from_args(self.args, L.ENV_HIGHLEVEL, self.num_envs, int(env_config.get("seed", 0)), auto_reset=False,
                                   arena_offset=int(env_config.get("arena_offset", 0)))
            backend = _GpuHierBackend(cfg, int(env_config.get("device", 0)), self.args, env_config.get("policy_dir"), env_config.get("pilot"),
                                      pilot_rows=env_config.get("pilot_rows", "variants"))
        self._init_protocol(backend, self.num_envs, range(1, self.args.num_agents + 1))
        self._eval = bool(getattr(self.args, "eval_info", False))
        self._last_eval = None

    def _obs_dicts(self, obs, envs):
        ids = self._ids
        if isinstance(envs, range) and len(ids) == 3:
            return [{1: x, 2: y, 3: z} for x, y, z in zip(obs[:, 0], obs[:, 1], obs[:, 2])]   # row views of the caller's fresh copy
        return [{i: obs[e, i - 1] for i in ids} for e in envs]

    def _put_action(self, a, e, k, v):
        if k <= len(self._ids):
            v = int(v)
            if not 0 <= v <= 2:   # Discrete(3): an int8 cast would wrap 256 -> 0 silently
                raise ValueError(f"commander action {v} of agent {k} in sub-environment {e} is outside Discrete(3)")
            a[e, k - 1] = v

    def _pack_all(self, a, dicts):
        n, ids = len(dicts), self._ids
        if not all(len(d) == len(ids) for d in dicts):   # a dict with other agent ids: the per-row path decides what it means
            return False
        cols = [np.fromiter((d[i] for d in dicts), dtype=np.int64, count=n) for i in ids]   # Discrete(3) per agent id
        if any(c.min() < 0 or c.max() > 2 for c in cols):
            return False   # the per-row path names the offender
        a[:] = 0
        for i, c in zip(ids, cols):
            a[:, i - 1] = c
        return True

    def send_actions(self, action_dict):
        self._last_eval = None
        super().send_actions(action_dict)

    def _info_of(self, e):
        if not self._eval:
            return {}
        if self._last_eval is None:
            self._last_eval = self.b.eval_info()
        return {k: int(self._last_eval[e, i]) for i, k in enumerate(L.EVAL_KEYS)}
