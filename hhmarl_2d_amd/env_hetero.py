"""Drop-in for envs/env_hetero.py LowLevelEnv (reference boundary: envs/env_hetero.py:20-63,
envs/env_base.py:62-109) on top of the MI355X world.

Same constructor (`LowLevelEnv(env_config)` with env_config["args"] = the reference's argparse
Namespace), same attributes (observation_space, action_space, _agent_ids, the *_preferred_format
flags, _skip_env_checking), same `reset(*, seed=None, options=None) -> (obs, {})` and
`step(action_dict) -> (obs, rewards, terminateds, truncateds, infos)` dict protocol:
obs has every agent id (zeros when dead), rewards only the ids alive at step start,
`terminateds is truncateds == {"__all__": done}`.

Extra keys of env_config (all optional, none changes the reference semantics):
  num_envs   number of arenas held on the GPU (default 1).  With num_envs > 1 the dict values
             carry a leading arena axis; RLlib-style single-env callers leave it at 1.
  seed       keyed-RNG seed (the reference is unseeded: env_base.py:62-77 ignores `seed`)
  arena_offset  global id of arena 0 in the keyed RNG (default 0): workers that should not replay each other's episodes take disjoint ranges
  device     GPU index
The batched tensor API for native rollout drivers is `self.world` (hhmarl_2d_amd.world.World).
"""
import numpy as np
import torch

from . import _lib as L
from . import spaces
from .world import World, make_config

ACTION_DIM_AC1, ACTION_DIM_AC2 = 4, 3
OBS_AC1, OBS_AC2, OBS_ESC_AC1, OBS_ESC_AC2 = 26, 24, 30, 29

try:  # subclass RLlib's base class only when ray is importable (it is not in the build image)
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base  # pragma: no cover
except Exception:  # noqa: BLE001
    class _Base:
        def __init__(self):
            pass


def config_from_args(args, env_kind, num_envs, seed, auto_reset=False, arena_offset=0):
    return make_config(
        n_arenas=num_envs, env_kind=env_kind, level=args.level,
        agent_mode=L.MODE_FIGHT if args.agent_mode == "fight" else L.MODE_ESCAPE,
        n_agents=args.num_agents, n_opps=args.num_opps, horizon=args.horizon,
        friendly_kill=args.friendly_kill, friendly_punish=args.friendly_punish, esc_dist_rew=args.esc_dist_rew,
        hier_action_assess=getattr(args, "hier_action_assess", True),
        hier_opp_fight_ratio=getattr(args, "hier_opp_fight_ratio", 75), auto_reset=auto_reset,
        ext_opp_actions=(env_kind == L.ENV_LOWLEVEL and args.level >= 4), map_size=args.map_size,
        glob_frac=args.glob_frac, rew_scale=float(args.rew_scale), seed=seed, arena_offset=arena_offset,
        # evaluation.py's low-level-vs-low-level mode: the opponents fly L{eval_level_opp} fight policies (env_base.py:343-346,387-390)
        opp_side_selector=(env_kind == L.ENV_HIGHLEVEL and not getattr(args, "eval_hl", True)))


def _enable_trace(env):
    """env_config["record_trace"]: the world keeps a trajectory ring buffer of arena 0 ON THE DEVICE (hh_trace_enable; the role of
    cmano_simulator.py:125-130,159-162 record_unit_trace) — one row per unit after every reset and tick, read back only by plot()"""
    if env.record_trace:
        env.world.trace_enable(1, capacity=int(env.args.horizon) + 40)


def plot_trace(env, out_file, paths=True):
    if not env.record_trace:
        raise RuntimeError("plot() needs env_config['record_trace'] = True")
    rows, ep = env.world.trace_read()[0]
    if len(rows) == 0:
        raise RuntimeError("plot() needs at least one reset()")
    rows = rows[ep == ep[-1]]                         # the current episode of arena 0
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    nA, m = env.args.num_agents, env.args.map_size
    fig, ax = plt.subplots(figsize=(6, 6), dpi=120)
    pos, alive = rows[:, :, :2], rows[:, :, 4]        # [T, A, (lat, lon)], [T, A]
    for i in range(pos.shape[1]):
        col = "tab:blue" if i < nA else "tab:red"
        t_end = int(np.max(np.nonzero(alive[:, i])[0])) if alive[:, i].any() else 0
        if paths:
            ax.plot(pos[: t_end + 1, i, 1], pos[: t_end + 1, i, 0], "--", color=col, lw=1)
        mk = "^" if alive[-1, i] else "x"
        ax.plot(pos[t_end, i, 1], pos[t_end, i, 0], mk, color=col, ms=8)
        ax.annotate(f"r_{i + 1}", (pos[t_end, i, 1], pos[t_end, i, 0]), fontsize=8)
    for i in range(rows.shape[1]):
        if rows[-1, i, 7]:
            ax.plot(rows[-1, i, 6], rows[-1, i, 5], "*", color="tab:blue" if i < nA else "tab:red", ms=6)
    ax.set_xlim(7.0, 7.0 + m); ax.set_ylim(5.0, 5.0 + m)
    ax.set_xlabel("lon [deg E]"); ax.set_ylabel("lat [deg N]"); ax.set_title(f"arena 0, tick {len(rows) - 1}")
    if out_file is not None:
        fig.savefig(str(out_file))
    plt.close(fig)
    return out_file


class LowLevelEnv(_Base):
    """Low-Level Environment for Aircombat Maneuvering (2-vs-2), MI355X-resident."""

    def __init__(self, env_config):
        self.args = env_config.get("args", None)
        self.agent_mode = self.args.agent_mode
        self.obs_fight = {1: OBS_AC1, 2: OBS_AC2, 3: OBS_AC1, 4: OBS_AC2}
        self.obs_esc = {1: OBS_ESC_AC1, 2: OBS_ESC_AC2, 3: OBS_ESC_AC1, 4: OBS_ESC_AC2}
        self.obs_dim_map = self.obs_fight if self.agent_mode == "fight" else self.obs_esc
        self._obs_space_in_preferred_format = True
        self.observation_space = spaces.Dict({
            i: spaces.Box(low=np.zeros(self.obs_dim_map[i]), high=np.ones(self.obs_dim_map[i]), dtype=np.float32)
            for i in range(1, 5)})
        self._action_space_in_preferred_format = True
        self.action_space = spaces.Dict({
            1: spaces.MultiDiscrete([13, 9, 2, 2]), 2: spaces.MultiDiscrete([13, 9, 2]),
            3: spaces.MultiDiscrete([13, 9, 2, 2]), 4: spaces.MultiDiscrete([13, 9, 2])})
        self._agent_ids = set(range(1, self.args.num_agents + 1))
        self._skip_env_checking = True
        self.map_size = self.args.map_size
        self.num_envs = int(env_config.get("num_envs", 1))
        self.opponent_policy = env_config.get("opponent_policy", None)
        # env_hetero.py:23,50,55-59: "fight", except that level 5 in fight mode draws k = randint(3,5) at every reset():
        # the opponents fly policies[k] and observe in escape mode when k == 5.  The draw is the world's keyed one; the
        # facade mirrors it in opp_k / opp_mode (arrays over arenas when num_envs > 1) for the opponent_policy callable.
        self.opp_mode = "fight"
        self.opp_k = None
        self._l5_draw = self.args.level == 5 and self.agent_mode == "fight"
        policy_dir = env_config.get("policy_dir", None)
        if self.args.level >= 4 and self.opponent_policy is None and policy_dir is None:
            raise ValueError("levels 4-5 fly frozen opponent policies (envs/env_base.py:312-398); pass env_config['policy_dir'] = the "
                             "directory of the exported L*_AC*_{fight,escape}.pt files, or env_config['opponent_policy'] = "
                             "callable(opp_obs f32 [N,2,30], env) -> int8 actions [N,2,4] for units 3,4")
        cfg = config_from_args(self.args, L.ENV_LOWLEVEL, self.num_envs, int(env_config.get("seed", 0)), arena_offset=int(env_config.get("arena_offset", 0)))
        self.world = World(cfg, device=int(env_config.get("device", 0)))
        if self.args.level >= 4 and self.opponent_policy is None:   # _get_policies("LowLevel"), env_base.py:312-332
            from .pilots import OpponentNets, PolicyBank
            bank = PolicyBank.from_reference_dir(self.world.device, policy_dir, "LowLevel", self.args, max_rows=self.num_envs * 2)
            self.opponent_policy = OpponentNets(self.world, bank=bank, bind=True, skip_first=False)   # bound before any step_begin
        self._act = torch.zeros((self.num_envs, self.world.n_ctrl, 4), dtype=torch.int8, device=self.world.device)
        self._out = self.world.alloc_outputs()
        # pinned host mirrors: one asynchronous copy per array and ONE stream synchronisation per step() instead of a blocking
        # .cpu() per array (the dict protocol hands host arrays both ways on every call)
        self._act_pin = torch.zeros((self.num_envs, self.world.n_ctrl, 4), dtype=torch.int8).pin_memory()
        self._act_host = self._act_pin.numpy()
        self._out_pin = [torch.zeros(t.shape, dtype=t.dtype).pin_memory() for t in self._out]
        self.steps = 0
        self.rewards = {}
        self.record_trace = bool(env_config.get("record_trace", False))
        _enable_trace(self)
        super().__init__()

    # -- helpers
    def _to_host(self, outs):
        """device (obs, reward, valid, done) -> their pinned host mirrors as numpy arrays, one synchronisation"""
        for src, dst in zip(outs, self._out_pin):
            dst.copy_(src, non_blocking=True)
        torch.cuda.current_stream(self.world.device).synchronize()
        return [t.numpy() for t in self._out_pin]

    def _obs_dict(self, obs):
        o = obs if isinstance(obs, np.ndarray) else obs.cpu().numpy()
        if self.num_envs == 1:
            return {i: o[0, i - 1, : self.obs_dim_map[i]].copy() for i in sorted(self._agent_ids)}
        return {i: o[:, i - 1, : self.obs_dim_map[i]].copy() for i in sorted(self._agent_ids)}

    def _refresh_opp_policy(self):
        if self._l5_draw:
            k = self.world.opp_policy().cpu().numpy()
            modes = np.where(k == 5, "escape", "fight")
            self.opp_k, self.opp_mode = (int(k[0]), str(modes[0])) if self.num_envs == 1 else (k, modes)

    def reset(self, *, seed=None, options=None):
        self.steps = 0
        obs = self.world.reset()
        self._refresh_opp_policy()
        return self._obs_dict(obs), {}

    def state(self):
        return self._obs_dict(self.world.observe())

    def step(self, action):
        self.rewards = {}
        n_ag = self.args.num_agents
        if action:
            a = self._act_host
            a[:] = 0
            for k, v in action.items():
                v = np.asarray(v)
                if self.num_envs == 1:
                    a[0, k - 1, : v.shape[-1]] = v
                else:
                    a[:, k - 1, : v.shape[-1]] = v
            self._act.copy_(self._act_pin, non_blocking=True)
            if self.opponent_policy is not None:
                # env_hetero.py:160-172: agents act, then each frozen-policy opponent observes and acts
                mode = L.OPP_MODE_EPISODE if self._l5_draw else (0 if self.opp_mode == "fight" else 1)
                opp_obs = self.world.step_begin(self._act[:, :n_ag].contiguous(), mode)
                opp_act = self.opponent_policy(opp_obs, self).to(torch.int8).contiguous()
                obs, rew, val, done = self.world.step_finish(opp_act, out=self._out)
            else:
                obs, rew, val, done = self.world.step(self._act, out=self._out)
            self.steps += 1
            obs, rew, val, done = self._to_host((obs, rew, val, done))
            if self.num_envs == 1:
                self.rewards = {i: float(rew[0, i - 1]) for i in range(1, n_ag + 1) if val[0, i - 1]}
                d = bool(done[0])
            else:
                self.rewards = {i: np.where(val[:, i - 1] > 0, rew[:, i - 1], 0.0) for i in range(1, n_ag + 1)}
                d = done.astype(bool)
            obs_d = self._obs_dict(obs)
        else:  # the reference skips _take_action for an empty action dict (env_base.py:87-88)
            obs_d = self.state()
            dn = self.world.arena_status()[:, 3].cpu().numpy().astype(bool)   # 16 bytes per arena, not the world
            d = bool(dn[0]) if self.num_envs == 1 else dn
        terminateds = truncateds = {"__all__": d}
        return obs_d, self.rewards, terminateds, truncateds, {}

    def plot(self, out_file=None, paths=True):
        """Trajectory plot of arena 0 (the role of env_base.py:622-645 + warsim/scenplotter, without the cartopy
        background): needs env_config["record_trace"] = True (the world then keeps arena 0's trajectory in a ring buffer on the device)."""
        return plot_trace(self, out_file, paths)

    def close(self):
        self.world.close()
