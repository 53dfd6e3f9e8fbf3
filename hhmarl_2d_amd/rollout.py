"""Rollout post-processing next to the environment: the centralised-critic input
packing of train_hetero.py:113-181 and GAE (train_hetero.py:216) on device tensors."""
import ctypes as C

import torch

from . import _lib as L

ACTION_DIM_AC1, ACTION_DIM_AC2 = 4, 3


def _p(t):
    return C.c_void_p(t.data_ptr())


def gae(reward, value, valid, done, gamma=0.99, lam=0.95):
    """reward, valid [T,N,nA]; value [T+1,N,nA]; done [T,N] (device tensors) -> (advantages, returns)"""
    T, N, nA = reward.shape
    assert value.shape == (T + 1, N, nA) and valid.shape == reward.shape and done.shape == (T, N)
    reward, value = reward.contiguous().float(), value.contiguous().float()
    valid, done = valid.contiguous().to(torch.uint8), done.contiguous().to(torch.uint8)
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    st = C.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream)
    L.check(L.lib().hh_gae(T, N, nA, _p(reward), _p(value), _p(valid), _p(done), float(gamma), float(lam), _p(adv), _p(ret), st))
    return adv, ret


def gae_rllib(reward, value, done, gamma=0.99, lam=0.95):
    """Advantages / value targets the way RLlib 2.4 computes them for the reference's step stream (hh_gae_rllib): rows of agents
    without a reward key stay in with reward 0.0 (nothing is masked), last_r = 0.0 at every episode end, float64 discounted sum.
    train_hetero.py:216 uses lam = 0.95, train_hier.py:186 RLlib's default lam = 1.0.
    reward [T,N,nA] (0.0 where the world reported reward_valid = 0); value [T+1,N,nA]; done [T,N] -> (advantages, value targets)"""
    T, N, nA = reward.shape
    assert value.shape == (T + 1, N, nA) and done.shape == (T, N)
    reward, value = reward.contiguous().float(), value.contiguous().float()
    done = done.contiguous().to(torch.uint8)
    adv, ret = torch.empty_like(reward), torch.empty_like(reward)
    st = C.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream)
    L.check(L.lib().hh_gae_rllib(T, N, nA, _p(reward), _p(value), _p(done), float(gamma), float(lam), _p(adv), _p(ret), st))
    return adv, ret


def central_critic_inputs(obs, actions):
    """The four blocks the reference's centralised critic sees per agent (central_critic_observer +
    on_postprocess_trajectory, train_hetero.py:113-181), for the 2-vs-2 low-level setting:
      agent 1: obs_1_own = obs[1] (26|30), obs_2 = obs[2] (24|29), act_1_own = own action (4), act_2 = friend's (3)
      agent 2: obs_1_own = obs[2],         obs_2 = obs[1],          act_1_own = own (3),        act_2 = friend's (4)
    with the heading / speed components scaled by 1/12 and 1/8 (train_hetero.py:143-160).
    obs [..., 2, D] (zero padded rows as the world emits them), actions int8 [..., 2, 4].
    Returns a dict per agent of float32 tensors with the reference's widths."""
    d1 = obs.shape[-1]
    d2 = d1 - 2 if d1 == 26 else d1 - 1          # 26/24 fight, 30/29 escape
    # the reference divides the integer actions in float64 and stores into the float32 batch (train_hetero.py:143-160): the same here
    # on any device (a float32 division on the GPU is not correctly rounded in every PyTorch build, the float64 one is)
    a = actions.double()
    scaled = torch.stack([a[..., 0] / 12.0, a[..., 1] / 8.0, a[..., 2], a[..., 3]], dim=-1).float()
    o1, o2 = obs[..., 0, :d1], obs[..., 1, :d2]
    a1, a2 = scaled[..., 0, :ACTION_DIM_AC1], scaled[..., 1, :ACTION_DIM_AC2]
    return {1: {"obs_1_own": o1, "obs_2": o2, "act_1_own": a1, "act_2": a2},
            2: {"obs_1_own": o2, "obs_2": o1, "act_1_own": a2, "act_2": a1}}


def central_critic_rows(obs, actions, agent):
    """The flattened CUR_OBS rows the reference's critic is trained on, for the 2-vs-2 low-level setting: RLlib flattens the
    observer's Dict in sorted key order (act_1_own, act_2, obs_1_own, obs_2) and `on_postprocess_trajectory`
    (train_hetero.py:120-160) then writes the own and the friend's actions into the first 7 columns, heading / speed components
    scaled by 1/12 and 1/8.  Agent 1 (type 1): [own act 4 | friend act 3 | own obs 26|30 | friend obs 24|29]; agent 2 (type 2):
    [own act 3 | friend act 4 | own obs 24|29 | friend obs 26|30].  obs [..., 2, D] as the world emits it (zero padded), actions
    int8 [..., 2, 4] -> float32 [..., 57] (fight) / [..., 66] (escape).  Pinned by tests/golden/critic_packing.npz (recorded
    from the reference's own callbacks)."""
    c = central_critic_inputs(obs, actions)[agent]
    return torch.cat([c["act_1_own"], c["act_2"], c["obs_1_own"], c["obs_2"]], dim=-1).float()


def central_critic_rows_hl(obs, actions, agent):
    """train_hier.py:100-165 for the 3-vs-3 commander: sorted keys (act_1_own, act_2, act_3, obs_1_own, obs_2, obs_3); the own
    action and the other two agents' actions (ascending id) divided by N_OPP_HL = 2 in columns 0..2.  obs [..., 3, 34],
    actions int8 [..., 3] -> float32 [..., 105]."""
    others = [i for i in (1, 2, 3) if i != agent]
    order = [agent] + others
    a = torch.stack([actions[..., i - 1].double() / 2.0 for i in order], dim=-1).float()
    return torch.cat([a] + [obs[..., i - 1, :].float() for i in order], dim=-1)


def episode_segments(done):
    """batch_mode="complete_episodes" (train_hetero.py:212) on the [T, N] done flags of a rollout with auto-reset:
    -> (segment id per row [T, N], starting at 0 per arena and increasing after every done row;
        complete [T, N] bool: the row belongs to an episode that also ENDS inside this rollout — the rows RLlib would train on;
        the trailing fragment of every arena is carried into the next rollout instead)"""
    d = done.to(torch.int64)
    seg = torch.cumsum(d, dim=0) - d                     # episodes finished strictly before this row
    n_done = d.sum(dim=0, keepdim=True)                  # episodes that end inside the rollout, per arena
    return seg, seg < n_done


class PPORollout:
    """What RLlib's rollout workers produce for train_hetero.py's PPO (train_hetero.py:206-243), for every arena of a `World` at once and
    without leaving the device: per tick the two trainable policies are sampled by `hh_policy_sample` (actor forward, Categorical draw
    per action component from the keyed RNG, its log-probability, and the centralised value branch on central_critic_observer's row —
    the other agent's observation, action inputs zero while sampling) and the world takes one `hh_step`; after T ticks the rewards and
    value predictions become advantages and value targets (gamma 0.99, lambda 0.95: train_hetero.py:216).  The 2 T + 2 launches of a
    collect are one HIP graph (4 T + 2 at curriculum levels 4-5, where the frozen opponents' networks run between the two halves of
    every step).

    semantics = "rllib" (default) — the batch RLlib 2.4 hands the reference's learner (ray's sampler + compute_advantages as restated in
    oracle/gae_ref.py; include/hh_abi.h: hh_gae_rllib):
      * every agent's trajectory spans the whole episode: the rows of an agent that died earlier STAY IN with reward 0.0 (the reference
        returns an observation for every agent id, zeros for dead ones, and a reward only for ids alive at step start; RLlib fills
        `rewards.get(agent_id, 0.0)`) — nothing is masked, `valid` is only reported;
      * last_r = 0.0 at every episode end (terminateds["__all__"] also at the horizon, env_base.py:108): the recursion is cut there and
        no value is bootstrapped across it; float64 delta and discounted sum, float32 results;
      * train_hetero.py:212 batch_mode = "complete_episodes": RLlib trains on whole episodes only.  `complete` (bool [T, N]) marks the
        rows of episodes that END inside this collect — the rows of that batch; the trailing fragment of every arena (its episode is still
        running after tick T - 1) gets RLlib's truncated-trajectory bootstrap from one more value evaluation and is flagged
        complete = False, so that a learner either drops it or keeps it for the next batch (`segments` numbers the episodes per arena).
    semantics = "masked": the pre-round-5 convention (`hh_gae`): rows without a reward key have advantage = target = 0 and do not
    propagate, float32 throughout — for learners that cut dead agents' rows out.

    Buffers (device, overwritten by every `collect`):  obs f32 [T+1, N, 2, D] (row t = what the policy saw at tick t), actions i8
    [T, N, 2, 4], logp f32 [T, N, 2], vf f32 [T+1, N, 2], reward f32 [T, N, 2] (0.0 where valid = 0), valid u8 [T, N, 2], done u8
    [T, N], adv / target f32 [T, N, 2]; `complete` / `segments` are derived from `done` on demand.  `critic_rows(agent)` gives the
    flattened CUR_OBS rows the reference's critic is trained on (actions filled in the way on_postprocess_trajectory does)."""

    def __init__(self, world, bank, T, gamma=0.99, lam=0.95, use_graph=True, opponents=None, semantics="rllib"):
        """opponents: levels 4-5 only (env_hetero.py:160-172: frozen-policy opponents observe and act between the agents' actions and the tick) —
        a `pilots.OpponentNets(world, skip_first=False)` (its bank bound, so that hh_step_begin lists the opponents' rows itself) or any
        callable(opp_obs f32 [N, 2, 30] on the device, None) -> int8 [N, 2, 4] that only enqueues work on the current stream"""
        from . import pilots
        assert world.cfg.env_kind == L.ENV_LOWLEVEL and world.n_agents == 2 and world.cfg.auto_reset, "PPORollout drives an auto-resetting LowLevelEnv world"
        if semantics not in ("rllib", "masked"):
            raise ValueError("semantics: 'rllib' (RLlib 2.4's trajectory view, the reference's) or 'masked' (rows without a reward key cut out)")
        self.semantics = semantics
        self.split = bool(world.cfg.ext_opp_actions)
        if self.split and opponents is None:
            raise ValueError("levels 4-5 fly frozen opponent policies (envs/env_base.py:312-398): pass opponents = pilots.OpponentNets(world, skip_first=False)")
        if getattr(opponents, "_skip", 0):
            # OpponentNets(skip_first=True) serves a facade whose reset() already ran one step_begin: its one-shot path re-bins rows that
            # hh_step_begin listed, and captured into the collect's graph it would do so on every replay (every row evaluated twice).  The
            # object may be shared with such a facade, so it is refused rather than silently changed
            raise ValueError("PPORollout needs opponents = pilots.OpponentNets(world, skip_first=False): this one still holds the one-shot skip of a "
                             "facade-style binding (skip_first=True)")
        self.opponents = opponents
        self.w, self.bank, self.T, self.gamma, self.lam = world, bank, int(T), float(gamma), float(lam)
        N, D, dev = world.N, world.D, world.device
        esc = world.cfg.agent_mode == L.MODE_ESCAPE
        self.sel = torch.tensor([pilots.SEL_ESC1 if esc else pilots.SEL_FIGHT1, pilots.SEL_ESC2 if esc else pilots.SEL_FIGHT2],
                                dtype=torch.uint8, device=dev).repeat(N, 1).contiguous()
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.obs = z((self.T + 1, N, 2, D), torch.float32)
        self.actions = z((self.T, N, 2, 4), torch.int8)
        self.logp = z((self.T, N, 2), torch.float32)
        self.vf = z((self.T + 1, N, 2), torch.float32)
        self.reward = z((self.T, N, 2), torch.float32)
        self.valid = z((self.T, N, 2), torch.uint8)
        self.done = z((self.T, N), torch.uint8)
        self.adv = z((self.T, N, 2), torch.float32)
        self.target = z((self.T, N, 2), torch.float32)
        self._tmp_act, self._tmp_logp = z((N, 2, 4), torch.int8), z((N, 2), torch.float32)
        self._opp_obs = z((N, world.A - 2, 30), torch.float32) if self.split else None
        # level 5 in fight mode: every arena observes in the mode of its own episode's draw (HH_OPP_MODE_EPISODE); otherwise fight mode
        self._opp_mode = L.OPP_MODE_EPISODE if (world.cfg.level == 5 and world.cfg.agent_mode == L.MODE_FIGHT) else 0
        self.use_graph = use_graph
        self._graph = None
        self._started = False

    def start(self):
        """reset every arena; the first observation becomes row 0 of the next collect.  Also builds the row lists of the fixed agent ->
        network mapping (a greedy evaluation whose results are discarded), so that every later call re-uses them (sel = NULL)."""
        self.w.reset(obs=self.obs[self.T])
        self.bank.sample(self.obs[self.T], self.sel, greedy=True, actions=self._tmp_act, logp=self._tmp_logp, vf=self.vf[self.T])
        self._started = True

    def _run(self):
        T = self.T
        self.obs[0].copy_(self.obs[T])      # where the previous collect (or start) left every arena
        for t in range(T):                  # every launch reads and writes its tick's rows of the [T, ...] buffers in place
            self.bank.sample(self.obs[t], None, world=self.w, actions=self.actions[t], logp=self.logp[t], vf=self.vf[t])
            out = (self.obs[t + 1], self.reward[t], self.valid[t], self.done[t])
            if self.split:   # agents act -> the frozen opponents observe (the agents' same-tick weapon flags included) and act -> tick
                self.w.step_begin(self.actions[t], self._opp_mode, opp_obs=self._opp_obs)
                self.w.step_finish(self.opponents(self._opp_obs, None).contiguous(), out=out)
            else:
                self.w.step(self.actions[t], out=out)
        # value of the observation after the last tick: the bootstrap of every arena's unfinished tail (an arena that just finished starts a
        # new episode there: both recursions cut at done and never read it)
        self.bank.sample(self.obs[T], None, greedy=True, actions=self._tmp_act, logp=self._tmp_logp, vf=self.vf[T])
        st = C.c_void_p(torch.cuda.current_stream(self.w.device).cuda_stream)
        if self.semantics == "rllib":
            # the world writes reward 0.0 wherever it reports valid = 0 (a dead agent's row): exactly RLlib's rewards.get(agent_id, 0.0)
            L.check(L.lib().hh_gae_rllib(T, self.w.N, 2, _p(self.reward), _p(self.vf), _p(self.done), self.gamma, self.lam,
                                         _p(self.adv), _p(self.target), st))
        else:
            L.check(L.lib().hh_gae(T, self.w.N, 2, _p(self.reward), _p(self.vf), _p(self.valid), _p(self.done), self.gamma, self.lam,
                                   _p(self.adv), _p(self.target), st))

    def collect(self):
        """T ticks of every arena -> self (the buffers above): 2 T + 2 launches, replayed from ONE HIP graph; no host synchronisation."""
        if not self._started:
            self.start()
        if not self.use_graph:
            self._run()
        else:
            if self._graph is None:
                torch.cuda.synchronize(self.w.device)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._run()
            self._graph.replay()
        return self

    @property
    def segments(self):
        """int64 [T, N]: index of the episode a row belongs to, counted per arena from the start of this collect (episode_segments)"""
        return episode_segments(self.done)[0]

    @property
    def complete(self):
        """bool [T, N]: the row's episode ends inside this collect — batch_mode = "complete_episodes" (train_hetero.py:212) trains on these"""
        return episode_segments(self.done)[1]

    def critic_rows(self, agent):
        """the CUR_OBS rows of `agent` (1 | 2) for the T collected ticks with both agents' actions filled in (central_critic_rows)"""
        return central_critic_rows(self.obs[: self.T], self.actions, agent)
