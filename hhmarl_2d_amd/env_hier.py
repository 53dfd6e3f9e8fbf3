"""Drop-in for envs/env_hier.py HighLevelEnv (3-vs-3 commander) on top of the MI355X world.

Same surface as the reference (envs/env_hier.py:27-47): `HighLevelEnv(env_config)`,
`observation_space = Box(0,1,(34,))`, `action_space = Discrete(3)`, `_agent_ids = {1,2,3}`,
`reset() -> (obs, {})`, `step({1:a1, 2:a2, 3:a3}) -> (obs, rewards, terminateds, truncateds, {})`.
The frozen low-level pilots the reference loads inside the env (env_base.py:312-398, not shipped) are
supplied as `env_config["pilot"]` (see hhmarl_2d_amd/pilots.py); they run in PyTorch between the
phase kernels of the macro step (csrc/hh_kernels_hier.h)."""
import numpy as np
import torch

from . import _lib as L
from . import spaces
from .env_hetero import _Base, _enable_trace, config_from_args, plot_trace
from .world import World

N_OPP_HL = 2
OBS_HL = 14 + N_OPP_HL * 10
N_SUB_STEPS = 16  # while s <= self.n_sub_steps (15)


def macro_step(world, commander_actions, pilot, out=None, pilot_buf=None, early_exit=False):
    """One HighLevelEnv.step for every arena of `world` (env_hier.py:114-140).
    commander_actions: int8 [N, n_agents] on the world's device."""
    nA = world.n_agents
    if getattr(pilot, "variants", False):   # one launch + one policy call per sub-step (pilots.VariantNetPilot; same trajectories)
        po, pm = world.hl_begin_variants(commander_actions, pilot_buf)
        for sub in range(N_SUB_STEPS):
            po, pm, running = world.hl_act_tick(pilot(po, pm), pilot_buf, count_running=early_exit)
            if early_exit and running == 0:
                break
        return world.hl_end(out)
    po, pm = world.hl_begin(commander_actions, pilot_buf)
    for sub in range(N_SUB_STEPS):
        act = pilot(po, pm).contiguous()
        po, pm = world.hl_agents_act(act, pilot_buf)
        act_o = pilot(po, pm)
        if act_o.data_ptr() != act.data_ptr():
            act[:, nA:] = act_o[:, nA:]
        po, pm, running = world.hl_tick(act, pilot_buf, count_running=early_exit)
        if early_exit and running == 0:
            break
    return world.hl_end(out)


class HighLevelEnv(_Base):
    """High-Level Environment for Aircombat Maneuvering (commander), MI355X-resident."""

    def __init__(self, env_config):
        self.args = env_config.get("args", None)
        self.n_sub_steps = 15
        self.min_sub_steps = 10
        self.observation_space = spaces.Box(low=np.zeros(OBS_HL), high=np.ones(OBS_HL), dtype=np.float32)
        self.action_space = spaces.Discrete(N_OPP_HL + 1)
        self._agent_ids = set(range(1, self.args.num_agents + 1))
        self._skip_env_checking = True
        self.map_size = self.args.map_size
        self.num_envs = int(env_config.get("num_envs", 1))
        self.pilot = env_config.get("pilot", None)
        policy_dir = env_config.get("policy_dir", None)
        if self.pilot is None and policy_dir is None:
            raise ValueError("HighLevelEnv flies frozen low-level pilot policies (envs/env_base.py:312-398); pass env_config['policy_dir'] "
                             "= the directory of the exported L*_AC*_{fight,escape}.pt files, or env_config['pilot'] = "
                             "callable(pilot_obs, pilot_mode) -> int8 actions [N, A, 4] (A = 6 unit slots, 10 with more than three aircraft on a side)")
        cfg = config_from_args(self.args, L.ENV_HIGHLEVEL, self.num_envs, int(env_config.get("seed", 0)), arena_offset=int(env_config.get("arena_offset", 0)))
        self.world = World(cfg, device=int(env_config.get("device", 0)))
        if self.pilot is None:   # _get_policies("HighLevel"), env_base.py:333-343
            from .pilots import own_pilot
            self.pilot = own_pilot(self.world, policy_dir, self.args, env_config.get("pilot_rows", "variants"))
        self._cmd = torch.zeros((self.num_envs, self.args.num_agents), dtype=torch.int8, device=self.world.device)
        self.commander_actions = None
        self.rewards = {}
        self.record_trace = bool(env_config.get("record_trace", False))
        _enable_trace(self)
        super().__init__()

    def _obs_dict(self, obs):
        o = obs.cpu().numpy()
        if self.num_envs == 1:
            return {i: o[0, i - 1].copy() for i in sorted(self._agent_ids)}
        return {i: o[:, i - 1].copy() for i in sorted(self._agent_ids)}

    def reset(self, *, seed=None, options=None):
        self.commander_actions = None
        obs = self.world.reset()
        return self._obs_dict(obs), {}

    def state(self):
        return self._obs_dict(self.world.observe())

    def _eval_info(self):
        """env_base.py:91-107: win/lose/draw flags and fight / escape / target-choice counters over the units that still exist
        after the step (eval mode of evaluation.py:66-82), counted on the device inside hh_hl_end for every arena
        (hh_eval_info): ints for one arena, int32 arrays over arenas for num_envs > 1.  `self.world.eval_info()[1]` holds
        the sums over all commander steps — what evaluation.py accumulates into eval_stats."""
        last = self.world.eval_info()[0].cpu().numpy()
        if self.num_envs == 1:
            return {k: int(last[0, i]) for i, k in enumerate(L.EVAL_KEYS)}
        return {k: last[:, i].copy() for i, k in enumerate(L.EVAL_KEYS)}

    def _macro_step_batched(self):
        """large batches: no early exit (some arena is practically always still inside its macro step), and with the library's
        own NetPilot / VariantNetPilot — launches only, nothing the host has to see in between — the 66 (34) launches of a commander step are captured
        once in a HIP graph and replayed"""
        from .pilots import NetPilot, VariantNetPilot
        if not isinstance(self.pilot, (NetPilot, VariantNetPilot)):
            return macro_step(self.world, self._cmd, self.pilot, early_exit=False)
        # the captured graph holds the world's device pointers (trace ring, bound bank's row lists) and the bank's (weight blobs, selector table, the
        # kernel instance its tile width picks) by value: re-capture whenever World.trace_enable / bind_policy or PolicyBank.set_net / set_critic /
        # set_lut / set_tile_rows / close changed them since — the bank may be the caller's own and shared with other worlds
        gen = (getattr(self.world, "ptr_generation", 0), id(self.pilot.bank), getattr(self.pilot.bank, "generation", 0))
        if getattr(self, "_graph", None) is not None and self._graph_gen != gen:
            self._graph = None
        if getattr(self, "_graph", None) is None:
            self._g_out = self.world.alloc_outputs()
            self._g_pilot = self.world.alloc_pilot_variants() if getattr(self.pilot, "variants", False) else self.world.alloc_pilot()
            # the forward kernels' first launch must not happen inside a capture: one call on rows without a network
            nw = min(64, self.pilot.bank.max_rows)
            self.pilot.bank.act(torch.zeros((nw, 30), device=self.world.device), torch.zeros((nw,), dtype=torch.uint8, device=self.world.device))
            torch.cuda.synchronize(self.world.device)
            side = torch.cuda.Stream(device=self.world.device)
            side.wait_stream(torch.cuda.current_stream(self.world.device))
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    macro_step(self.world, self._cmd, self.pilot, out=self._g_out, pilot_buf=self._g_pilot, early_exit=False)
            torch.cuda.current_stream(self.world.device).wait_stream(side)
            self._graph = graph
            self._graph_gen = gen
        self._graph.replay()
        return self._g_out

    def step(self, action):
        self.rewards = {}
        info = {}
        nA = self.args.num_agents
        if action:
            self.commander_actions = action
            c = np.zeros((self.num_envs, nA), dtype=np.int8)
            for k, v in action.items():
                if k <= nA:
                    c[:, k - 1] = np.asarray(v)
            self._cmd.copy_(torch.from_numpy(c))
            # the library's own pilot: the commander step's launches replayed from ONE HIP graph, all 16 sub-steps (arenas whose macro step is
            # over idle) — 0.65 against 1.03 ms for a single environment: leaving the loop early costs a host synchronisation per tick, more than the
            # ~1.5 sub-steps it saves.  A foreign pilot is called eagerly, with the early exit for a handful of arenas (it may be Python per row).
            from .pilots import NetPilot, VariantNetPilot
            if isinstance(self.pilot, (NetPilot, VariantNetPilot)):
                obs, rew, val, done = self._macro_step_batched()
            else:
                obs, rew, val, done = macro_step(self.world, self._cmd, self.pilot, early_exit=self.num_envs <= 64)
            rew, val, done = rew.cpu().numpy(), val.cpu().numpy(), done.cpu().numpy()
            if getattr(self.args, "eval_info", False):
                info = self._eval_info()
            if self.num_envs == 1:
                self.rewards = {i: float(rew[0, i - 1]) for i in range(1, nA + 1) if val[0, i - 1]}
                d = bool(done[0])
            else:
                self.rewards = {i: rew[:, i - 1] for i in range(1, nA + 1)}
                d = done.astype(bool)
            obs_d = self._obs_dict(obs)
        else:
            obs_d = self.state()
            dn = self.world.arena_status()[:, 3].cpu().numpy().astype(bool)
            d = bool(dn[0]) if self.num_envs == 1 else dn
        terminateds = truncateds = {"__all__": d}
        return obs_d, self.rewards, terminateds, truncateds, info

    def plot(self, out_file=None, paths=True):
        return plot_trace(self, out_file, paths)

    def close(self):
        self.world.close()
