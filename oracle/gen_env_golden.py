"""
TEST INFRASTRUCTURE — records golden step traces from the REAL reference environment.

Runs only in the build container (needs /root/reference); writes tests/golden/env_*.npz, which
are data (inputs + expected outputs) and travel to the GPU box.  The reference is imported
unchanged behind the stand-ins of oracle/ref_harness.py; its randomness is the keyed tape of
include/hh_rng.h; its geodesic is oracle/geodesic_ref.py (geographiclib is not installable here).

Record layout (R rows per file):
  kind[R]      0 = reset row (state after env.reset()), 1 = step row (state after env.step())
  actions[R,A,4]   MultiDiscrete actions fed to env.step (agents; zeros elsewhere / on reset rows)
  ac_f, ac_i, rk_f, rk_i, ar_i   world snapshot in hh_get_state layout (include/hh_abi.h)
  obs[R,nA,D] f32, reward[R,nA] f64, valid[R,nA] u8 (key present in the rewards dict), done[R] u8
  meta         json: config kwargs, seed, arena id

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_env_golden.py [--out DIR] [names...]     (re)generate
      PYTHONDONTWRITEBYTECODE=1 python oracle/gen_env_golden.py --check                      regenerate to a temp dir and
                                                                                            compare with tests/golden
Everything is deterministic (keyed tape, crc32-seeded action policies), so --check must report no difference;
tests/test_golden_provenance.py runs it whenever /root/reference is present.
"""
import json
import math
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def random_actions(rng, env):
    return {1: [int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2)), int(rng.integers(2))],
            2: [int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2))]}


def pursuit_actions(rng, env):
    """Steer every agent at its nearest opponent and keep shooting (exercises cannon/missile kills)."""
    e = env.env
    sim = e.sim
    acts = {}
    for i in range(1, e.args.num_agents + 1):
        n = 4 if i == 1 or (i > 2 and sim.unit_exists(i) and sim.get_unit(i).ac_type == 1) else 3
        a = [6, int(rng.integers(4, 9)), 1, int(rng.random() < 0.5)][:n]
        if sim.unit_exists(i):
            u = sim.get_unit(i)
            best = None
            for j in range(e.args.num_agents + 1, e.args.total_num + 1):
                if sim.unit_exists(j):
                    o = sim.get_unit(j)
                    d = math.hypot(o.position.lon - u.position.lon, o.position.lat - u.position.lat)
                    if best is None or d < best[0]:
                        best = (d, o)
            if best is not None:
                o = best[1]
                brg = math.degrees(math.atan2(o.position.lon - u.position.lon, o.position.lat - u.position.lat)) % 360
                rel = (brg - u.heading + 180) % 360 - 180
                a[0] = int(np.clip(round(rel / 15.0) + 6, 0, 12))
                if best[0] < 0.05:
                    a[1] = int(rng.integers(0, 4))
        acts[i] = a
    return acts


SCENARIOS = [
    # name, kind, args kwargs, action policy, episodes, max rows
    ("l1_fight_random", "low", dict(level=1), random_actions, 2, 260),
    ("l1_fight_pursuit", "low", dict(level=1), pursuit_actions, 3, 300),
    ("l2_fight_pursuit", "low", dict(level=2), pursuit_actions, 3, 320),
    ("l3_fight_random", "low", dict(level=3), random_actions, 2, 320),
    ("l3_fight_pursuit_share", "low", dict(level=3, glob_frac=0.5, friendly_punish=True, rew_scale=2), pursuit_actions, 4, 420),
    ("l3_escape_shaping", "low", dict(level=3, agent_mode="escape", esc_dist_rew=True), pursuit_actions, 3, 360),
    ("l3_fight_nofriendly", "low", dict(level=3, friendly_kill=False), pursuit_actions, 3, 320),
    # levels 4-5: opponents fly frozen policies (env_base.py:312-398, files not shipped): taped actions, and the
    # opponents' own observations (the reference's lowlevel_state, evaluated after the agents acted) are recorded
    ("l4_fight_frozen_opps", "low", dict(level=4), pursuit_actions, 3, 300),
    ("l5_fight_frozen_opps", "low", dict(level=5, horizon=70), pursuit_actions, 6, 420),   # short horizon: several episodes, i.e. several draws of k (env_hetero.py:57)
]


def record(name, kind, kw, policy, episodes, max_rows, seed=20240917, arena=7):
    args = H.make_args(**kw)
    ext = kind == "low" and args.level >= 4
    opp_tape = {}
    if ext:
        ref = H.load_reference()
        cls = ref["env_hetero"].LowLevelEnv

        def get_policies(self_, mode):
            self_.policy = None
            self_.policies = {3: None, 4: None, 5: None}

        def policy_actions(self_, policy_type, agent_id, unit):
            st = self_.lowlevel_state(policy_type, agent_id, unit=unit)[agent_id]
            u = self_.sim.get_unit(agent_id)
            a = [int(opp_tape["rng"].integers(13)), int(opp_tape["rng"].integers(9)), 1, int(opp_tape["rng"].random() < 0.5)]
            tgt = self_.opp_to_attack[agent_id]
            if tgt and opp_tape["rng"].random() < 0.8:
                o = self_.sim.get_unit(tgt)
                brg = math.degrees(math.atan2(o.position.lon - u.position.lon, o.position.lat - u.position.lat)) % 360
                rel = (brg - u.heading + 180) % 360 - 180
                a[0] = int(np.clip(round(rel / 15.0) + 6, 0, 12))
            opp_tape["obs"][agent_id] = np.asarray(st, dtype=np.float32)
            opp_tape["act"][agent_id] = a
            opp_tape["mode"] = 0 if policy_type == "fight" else 1
            return {agent_id: np.array(a)}

        cls._get_policies = get_policies
        cls._policy_actions = policy_actions
    env = H.RefEnv(kind, args, seed=seed, arena=arena)
    A, nA = args.total_num, args.num_agents
    D = (26 if args.agent_mode == "fight" else 30) if kind == "low" else 34
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    rows = dict(kind=[], actions=[], ac_f=[], ac_i=[], rk_f=[], rk_i=[], ar_i=[], obs=[], reward=[], valid=[], done=[],
                opp_obs=[], opp_mode=[])
    opp_tape.update(rng=np.random.default_rng(zlib.crc32((name + "/opp").encode())), obs={}, act={}, mode=0)

    def push(k, act, obs, rew, done):
        st = env.state()
        rows["kind"].append(k)
        a = np.zeros((A, 4), dtype=np.int8)
        for i, v in (act or {}).items():
            a[i - 1, : len(v)] = v
        oo = np.zeros((A - nA, 30), dtype=np.float32)
        if k == 1:
            for i, v in opp_tape["act"].items():
                a[i - 1, : len(v)] = v
            for i, v in opp_tape["obs"].items():
                oo[i - nA - 1, : len(v)] = v
        rows["opp_obs"].append(oo)
        rows["opp_mode"].append(opp_tape["mode"])
        opp_tape["obs"], opp_tape["act"] = {}, {}
        rows["actions"].append(a)
        for key in ("ac_f", "ac_i", "rk_i", "ar_i"):
            rows[key].append(st[key])
        rows["rk_f"].append(st["rk_f"][:, :4])
        rows["obs"].append(env.obs_array(obs, D))
        r = np.zeros(nA)
        v = np.zeros(nA, dtype=np.uint8)
        for i, x in (rew or {}).items():
            r[i - 1] = x
            v[i - 1] = 1
        rows["reward"].append(r)
        rows["valid"].append(v)
        rows["done"].append(int(done))

    stats = dict(cannon_kills=0, rocket_kills=0, launches=0, oob=0)
    for ep in range(episodes):
        obs = env.reset()
        push(0, None, obs, None, False)
        done = False
        while not done and len(rows["kind"]) < max_rows:
            act = policy(rng, env)
            alive_before = {i for i in range(1, A + 1) if env.env.sim.unit_exists(i)}
            obs, rew, term, trunc, info = env.step(act)
            done = term["__all__"]
            push(1, act, obs, rew, done)
        if len(rows["kind"]) >= max_rows:
            break
    meta = dict(name=name, env=kind, args={k: v for k, v in vars(args).items()}, seed=seed, arena=arena, obs_dim=D,
                draws=len(env.tape.log))
    assert len(set(env.tape.log)) == len(env.tape.log), "keyed-RNG key collision"
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"env_{name}.npz")
    np.savez_compressed(path, **out)
    kills = int((np.diff(out["ac_i"][:, :, 0].astype(int), axis=0) < 0).sum())
    print(f"{name}: rows={len(out['kind'])} draws={meta['draws']} deaths={kills} "
          f"rocket_rows={int(out['rk_i'][:, :, 0].any(axis=1).sum())} size={os.path.getsize(path)}")


# ---------------------------------------------------------------- HighLevelEnv (3-vs-3 commander)
HL_SCENARIOS = [
    # name, args kwargs, pilot style, episodes, max commander steps
    ("hl_random_pilots", dict(mode=1), "random", 2, 70),
    ("hl_pursuit_pilots", dict(mode=1), "pursuit", 4, 90),
    ("hl_pursuit_share", dict(mode=1, glob_frac=0.3, hier_opp_fight_ratio=50, hier_action_assess=False), "pursuit", 3, 70),
    ("hl_eval_info", dict(mode=1, eval_info=True), "pursuit", 3, 60),   # evaluation.py mode: info dict of env_base.py:91-107
    # evaluation.py's n-vs-m scenarios (README.md:43 of the reference): fewer than six aircraft, asymmetric sides
    ("hl_2v3_eval", dict(mode=1, num_agents=2, num_opps=3, eval_info=True, horizon=200), "pursuit", 3, 40),
    ("hl_3v1_eval", dict(mode=1, num_agents=3, num_opps=1, eval_info=True, horizon=200, hier_opp_fight_ratio=100), "pursuit", 3, 30),
]


def record_hl(name, kw, style, episodes, max_rows, seed=20240917, arena=11):
    """HighLevelEnv.step = _action_assess + <=16 x {pilot policy -> _take_base_action -> do_tick}.
    The frozen pilot networks are not shipped (.gitignore:5), so `_policy_actions` is replaced by a
    tape: it still evaluates the reference's own lowlevel_state (recorded = expected pilot
    observation) and returns a scripted action (recorded = input)."""
    args = H.make_args(**kw)
    env = H.RefEnv("high", args, seed=seed, arena=arena)
    A, nA = args.total_num, args.num_agents
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cls = type(env.env)
    sub = dict(obs=[], mode=[], act=[])
    cur = {"last": 99}

    def policy_actions(self_, policy_type, agent_id, unit):
        st = self_.lowlevel_state(policy_type, agent_id, unit=unit)[agent_id]
        if agent_id <= cur["last"]:  # first live unit of a new sub-step
            sub["obs"].append(np.zeros((A, 30), dtype=np.float32))
            sub["mode"].append(np.zeros(A, dtype=np.uint8))
            sub["act"].append(np.zeros((A, 4), dtype=np.int8))
        cur["last"] = agent_id
        a = [int(rng.integers(13)), int(rng.integers(9)), int(rng.integers(2)), int(rng.integers(2))]
        if style == "pursuit" and policy_type == "fight":
            tgt = self_.opp_to_attack[agent_id][self_.commander_actions[agent_id] - 1][0]
            o = self_.sim.get_unit(tgt)
            brg = math.degrees(math.atan2(o.position.lon - unit.position.lon, o.position.lat - unit.position.lat)) % 360
            rel = (brg - unit.heading + 180) % 360 - 180
            a = [int(np.clip(round(rel / 15.0) + 6, 0, 12)), int(rng.integers(3, 9)), 1, int(rng.random() < 0.5)]
        sub["obs"][-1][agent_id - 1, : len(st)] = st
        sub["mode"][-1][agent_id - 1] = 1 if policy_type == "fight" else 2
        sub["act"][-1][agent_id - 1] = a
        return {agent_id: np.array(a)}

    cls._policy_actions = policy_actions
    rows = dict(kind=[], cmd=[], nsub=[], ac_f=[], ac_i=[], rk_f=[], rk_i=[], ar_i=[], tgt_id=[], tgt_d=[], obs=[],
                reward=[], valid=[], done=[], cmd_all=[])
    infos = []

    def push(k, cmd, nsub, obs, rew, done, cmd_all):
        st = env.state()
        rows["kind"].append(k)
        rows["cmd"].append(cmd)
        rows["nsub"].append(nsub)
        for key in ("ac_f", "ac_i", "rk_i", "ar_i", "tgt_id", "tgt_d"):
            rows[key].append(st[key])
        rows["rk_f"].append(st["rk_f"][:, :4])
        rows["obs"].append(env.obs_array(obs, 34))
        r = np.zeros(nA)
        v = np.zeros(nA, dtype=np.uint8)
        for i, x in (rew or {}).items():
            r[i - 1] = x
            v[i - 1] = 1
        rows["reward"].append(r)
        rows["valid"].append(v)
        rows["done"].append(int(done))
        rows["cmd_all"].append(cmd_all)

    for ep in range(episodes):
        obs = env.reset()
        push(0, np.zeros(nA, dtype=np.int8), 0, obs, None, False, np.zeros(A, dtype=np.int8))
        infos.append({})
        done = False
        while not done and len(rows["kind"]) < max_rows:
            cmd = rng.integers(0, 3, nA).astype(np.int8)
            n0 = len(sub["obs"])
            cur["last"] = 99
            cd = {i + 1: int(cmd[i]) for i in range(nA)}
            obs, rew, term, trunc, info = env.step(cd)
            done = term["__all__"]
            ca = np.array([(cd.get(i) or 0) for i in range(1, A + 1)], dtype=np.int8)  # env expanded it in place
            push(1, cmd, len(sub["obs"]) - n0, obs, rew, done, ca)
            infos.append({k: int(v) for k, v in info.items()})
        if len(rows["kind"]) >= max_rows:
            break
    meta = dict(name=name, env="high", args={k: v for k, v in vars(args).items()}, seed=seed, arena=arena, obs_dim=34,
                draws=len(env.tape.log))
    assert len(set(env.tape.log)) == len(env.tape.log), "keyed-RNG key collision"
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["sub_obs"] = np.asarray(sub["obs"])
    out["sub_mode"] = np.asarray(sub["mode"])
    out["sub_act"] = np.asarray(sub["act"])
    out["infos"] = np.array(json.dumps(infos))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"env_{name}.npz")
    np.savez_compressed(path, **out)
    deaths = int((np.diff(out["ac_i"][:, :, 0].astype(int), axis=0) < 0).sum())
    print(f"{name}: rows={len(out['kind'])} substeps={len(sub['obs'])} mean_sub={out['nsub'][out['kind'] == 1].mean():.1f} "
          f"draws={meta['draws']} deaths={deaths} size={os.path.getsize(path)}")


# ---------------------------------------------------------------- small traces: every level x mode, reward options, n-vs-m
#  short recordings on OTHER (seed, arena) pairs than the scenarios above, so that the GPU-side replay
# (tests/test_gpu_parity.py, tests/test_gpu_hier.py) pins the shared headers on the MI355X for every level / mode / reward option /
# side size, not only in the build container (oracle/soak_vs_reference.py replays 260 such traces through the oracle there).
MINI_SCENARIOS = [
    # name, kind, args kwargs, action policy, episodes, max rows, seed, arena
    ("fz_l1_escape", "low", dict(level=1, agent_mode="escape", horizon=60), random_actions, 2, 90, 777131, 1037),
    ("fz_l2_escape_shaping", "low", dict(level=2, agent_mode="escape", esc_dist_rew=True, horizon=80), pursuit_actions, 2, 90, 777262, 1074),
    ("fz_l2_fight_random", "low", dict(level=2, horizon=90, glob_frac=0.3), random_actions, 2, 90, 777393, 1111),
    ("fz_l1_share_punish", "low", dict(level=1, glob_frac=0.5, friendly_punish=True, rew_scale=2, horizon=70), pursuit_actions, 2, 90, 777524, 1148),
    ("fz_l2_nofriendly_map04", "low", dict(level=2, friendly_kill=False, map_size=0.4, horizon=100), pursuit_actions, 2, 100, 777655, 1185),
    ("fz_l3_escape_random", "low", dict(level=3, agent_mode="escape", horizon=80), random_actions, 2, 90, 777786, 1222),
    ("fz_l3_escape_share", "low", dict(level=3, agent_mode="escape", esc_dist_rew=True, glob_frac=0.3, rew_scale=2, horizon=90), pursuit_actions, 2, 90, 777917, 1259),
    ("fz_l3_fight_map04", "low", dict(level=3, map_size=0.4, horizon=120), pursuit_actions, 2, 120, 778048, 1296),
    ("fz_l3_fight_punish", "low", dict(level=3, friendly_punish=True, glob_frac=0.5, horizon=100), pursuit_actions, 2, 100, 778179, 1333),
    ("fz_l3_fight_short", "low", dict(level=3, horizon=40), random_actions, 3, 100, 778310, 1370),
    ("fz_l4_taped_share", "low", dict(level=4, glob_frac=0.3, horizon=80), pursuit_actions, 2, 90, 778441, 1407),
    ("fz_l4_taped_nofriendly", "low", dict(level=4, friendly_kill=False, rew_scale=2, horizon=80), random_actions, 2, 90, 778572, 1444),
    ("fz_l5_taped_punish", "low", dict(level=5, friendly_punish=True, horizon=50), pursuit_actions, 3, 110, 778703, 1481),
    ("fz_l5_escape_taped", "low", dict(level=5, agent_mode="escape", esc_dist_rew=True, horizon=70), pursuit_actions, 2, 90, 778834, 1518),
]
MINI_HL_SCENARIOS = [
    # name, args kwargs, pilot style, episodes, max commander steps, seed, arena
    ("hl_fz_1v1", dict(mode=1, num_agents=1, num_opps=1, horizon=150, eval_info=True), "pursuit", 2, 12, 555017, 3016),
    ("hl_fz_1v3_noassess", dict(mode=1, num_agents=1, num_opps=3, horizon=150, hier_action_assess=False, hier_opp_fight_ratio=100), "pursuit", 2, 12, 555034, 3027),
    ("hl_fz_3v2_share", dict(mode=1, num_agents=3, num_opps=2, horizon=150, glob_frac=0.3, hier_opp_fight_ratio=50, eval_info=True), "pursuit", 2, 12, 555051, 3038),
    ("hl_fz_2v2_escape_opps", dict(mode=1, num_agents=2, num_opps=2, horizon=120, hier_opp_fight_ratio=0), "random", 2, 12, 555068, 3049),
    ("hl_fz_3v3_nofriendly", dict(mode=1, friendly_kill=False, horizon=120, hier_opp_fight_ratio=100, eval_info=True), "pursuit", 2, 12, 555085, 3060),
    ("hl_fz_3v3_random_short", dict(mode=1, horizon=60, glob_frac=0.3, hier_action_assess=False), "random", 3, 12, 555102, 3071),
    ("hl_fz_2v1", dict(mode=1, num_agents=2, num_opps=1, horizon=150, hier_opp_fight_ratio=75, eval_info=True), "pursuit", 2, 12, 555119, 3082),
    # more than three aircraft on a side (README.md:43 "any n-vs-m"): ten unit slots, target lists of up to five (round 5)
    ("hl_fz_5v5_share", dict(mode=1, num_agents=5, num_opps=5, horizon=150, glob_frac=0.3, eval_info=True), "pursuit", 2, 14, 555136, 3093),
    ("hl_fz_4v4_random", dict(mode=1, num_agents=4, num_opps=4, horizon=100, hier_opp_fight_ratio=50, hier_action_assess=False), "random", 2, 12, 555153, 3104),
    ("hl_fz_5v2_fight", dict(mode=1, num_agents=5, num_opps=2, horizon=150, hier_opp_fight_ratio=100, eval_info=True), "pursuit", 2, 12, 555170, 3115),
    ("hl_fz_1v4_escape_opps", dict(mode=1, num_agents=1, num_opps=4, horizon=120, hier_opp_fight_ratio=0), "pursuit", 2, 12, 555187, 3126),
    ("hl_fz_4v5_nofriendly", dict(mode=1, num_agents=4, num_opps=5, horizon=150, friendly_kill=False, hier_opp_fight_ratio=100, eval_info=True), "pursuit", 2, 14, 555204, 3137),
]


# ---------------------------------------------------------------- the reference's OWN _get_policies / _policy_actions in the loop
# (VERDICT r2 "missing 1"): reference env -> lowlevel_state -> reference Fight1/Fight2/Esc1/Esc2.forward -> get_torch_action ->
# _take_base_action, recorded action for action.  Nothing of the reference's policy path is replaced: the only stand-in is
# `torch.load` (the exported policies/*.pt are not shipped, .gitignore:5), which hands out instances of the reference's own
# model classes carrying seeded synthetic weights (hhmarl_2d_amd.policy_nets.random_weights); `_get_policies` picks the files
# (env_base.py:312-347, incl. the L5 -> L3 escape fallback), `_policy_actions` (env_base.py:349-398) builds the input dict,
# calls forward() and takes the arg-max of the Categoricals.  A thin recorder around each model keeps the observation row the
# network was handed and its logits (-> top-2 margin per decision, so that a replay knows which arg-max was a near-tie).
NET_FILES = {   # file name -> (architecture, weight seed); every level / role carries its own weights so that a mix-up shows
    "L3_AC1_fight.pt": (0, 31), "L3_AC2_fight.pt": (1, 31), "L4_AC1_fight.pt": (0, 41), "L4_AC2_fight.pt": (1, 41),
    "L5_AC1_fight.pt": (0, 51), "L5_AC2_fight.pt": (1, 51), "L3_AC1_escape.pt": (2, 32), "L3_AC2_escape.pt": (3, 32),
    "L5_AC1_escape.pt": (2, 52), "L5_AC2_escape.pt": (3, 52),
}


class RecordingPolicy:
    """what torch.load returns in these recordings: the reference's model, called exactly as _policy_actions calls it"""

    def __init__(self, model, fname, log):
        self.model, self.fname, self.log = model, fname, log

    def __call__(self, input_dict, state, seq_lens):
        out = self.model(input_dict=input_dict, state=state, seq_lens=seq_lens)
        self.log.append((self.fname, np.asarray(input_dict["obs"]["obs_1_own"])[0].copy(), out[0][0].detach().numpy().copy()))
        return out


def reference_policy_loader(available, log):
    """stand-in for torch.load inside _get_policies: basename -> RecordingPolicy(reference model class with synthetic weights).
    A torch.load()ed policy is its own object graph: ac_models_hetero.py shares ONE module-level SHARED_LAYER between all
    instances it constructs, so every model is deep-copied after its weights went in (what unpickling a file gives)."""
    import copy
    import torch
    import gen_policy_golden as GP
    from hhmarl_2d_amd import policy_nets as PN
    M = GP.reference_models()
    classes = {PN.FIGHT1: M.Fight1, PN.FIGHT2: M.Fight2, PN.ESC1: M.Esc1, PN.ESC2: M.Esc2}
    cache = {}

    def load(path, *a, **k):
        name = os.path.basename(path)
        if name not in available:
            raise FileNotFoundError(path)
        if name not in cache:
            kind, seed = NET_FILES[name]
            model = classes[kind](None, None, PN.N_OUT[kind], {}, name)
            full = model.state_dict()
            for key, v in PN.random_weights(kind, seed).items():
                assert full[key].shape == v.shape, (key, full[key].shape, v.shape)
                full[key] = torch.from_numpy(v)
            model.load_state_dict(full)
            model = copy.deepcopy(model)
            model.eval()
            cache[name] = RecordingPolicy(model, name, log)
        return cache[name]
    return load


def decision_margin(logits, n_comp):
    """smallest top-2 logit gap over the MultiDiscrete components of one decision"""
    parts = np.split(logits, np.cumsum((13, 9, 2, 2)[:n_comp])[:-1])
    return float(min(np.sort(p)[-1] - np.sort(p)[-2] for p in parts))


def _restore_policy_path(cls):
    """earlier recorders of this module replace _get_policies / _policy_actions ON the env class: take the replacements off
    again so that the base class's (the reference's own) methods run"""
    for name in ("_get_policies", "_policy_actions"):
        if name in cls.__dict__:
            delattr(cls, name)


NET_SCENARIOS = [
    # name, args kwargs, files present in the policy directory, agents' action policy, episodes, max rows
    ("l4_fight_nets", dict(level=4), ("L3_AC1_fight.pt", "L3_AC2_fight.pt"), pursuit_actions, 3, 300),
    ("l5_fight_nets", dict(level=5, horizon=70), ("L3_AC1_fight.pt", "L3_AC2_fight.pt", "L4_AC1_fight.pt", "L4_AC2_fight.pt", "L3_AC1_escape.pt",
                                                  "L3_AC2_escape.pt"), pursuit_actions, 7, 480),   # seven episodes = seven draws of k
    ("l5_escape_nets", dict(level=5, agent_mode="escape", horizon=120), ("L5_AC1_fight.pt", "L5_AC2_fight.pt"), random_actions, 2, 200),
]


def record_nets(name, kw, files, policy, episodes, max_rows, seed=20240917, arena=7):
    """LowLevelEnv levels 4-5 with the reference's own frozen-opponent path running (env_hetero.py:48-59,160-172).  Same row
    layout as record() (so every replay of the taped traces also replays these), plus opp_margin [R, 2] (top-2 logit gap of
    each opponent's decision, inf where it took none) and opp_logits [R, 2, 26]; meta['policy_files'] names the weights."""
    import torch
    args = H.make_args(**kw)
    ref = H.load_reference()
    cls = ref["env_hetero"].LowLevelEnv
    _restore_policy_path(cls)
    log = []
    real_load = torch.load
    torch.load = reference_policy_loader(set(files), log)
    try:
        env = H.RefEnv("low", args, seed=seed, arena=arena)
    finally:
        torch.load = real_load
    A, nA = args.total_num, args.num_agents
    D = 26 if args.agent_mode == "fight" else 30
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    rows = dict(kind=[], actions=[], ac_f=[], ac_i=[], rk_f=[], rk_i=[], ar_i=[], obs=[], reward=[], valid=[], done=[],
                opp_obs=[], opp_mode=[], opp_margin=[], opp_logits=[], opp_file=[])
    names = sorted(NET_FILES)
    cur = {}
    base_pa = ref["env_base"].HHMARLBaseEnv._policy_actions

    def spy(self_, policy_type, agent_id, unit):   # records WHO decided (the log entry has no unit id); the decision is the reference's
        n0 = len(log)
        out = base_pa(self_, policy_type, agent_id, unit)
        assert len(log) == n0 + 1
        cur[agent_id] = (log[-1], np.asarray(out[agent_id]), policy_type)
        return out
    cls._policy_actions = spy

    def push(k, act, obs, rew, done):
        st = env.state()
        rows["kind"].append(k)
        a = np.zeros((A, 4), dtype=np.int8)
        for i, v in (act or {}).items():
            a[i - 1, : len(v)] = v
        oo = np.zeros((A - nA, 30), dtype=np.float32)
        mg = np.full((A - nA,), np.inf)
        lg = np.zeros((A - nA, 26), dtype=np.float32)
        fl = np.full((A - nA,), -1, dtype=np.int8)
        mode = 0 if env.env.opp_mode == "fight" else 1
        if k == 1:
            for i, ((fname, o, logits), action, ptype) in cur.items():
                a[i - 1, : len(action)] = action
                oo[i - nA - 1, : len(o)] = o
                lg[i - nA - 1, : len(logits)] = logits
                mg[i - nA - 1] = decision_margin(logits, len(action))
                fl[i - nA - 1] = names.index(fname)
        cur.clear()
        rows["opp_obs"].append(oo); rows["opp_margin"].append(mg); rows["opp_logits"].append(lg); rows["opp_file"].append(fl)
        rows["opp_mode"].append(mode_before["m"] if k == 1 else mode)
        rows["actions"].append(a)
        for key in ("ac_f", "ac_i", "rk_i", "ar_i"):
            rows[key].append(st[key])
        rows["rk_f"].append(st["rk_f"][:, :4])
        rows["obs"].append(env.obs_array(obs, D))
        r = np.zeros(nA)
        v = np.zeros(nA, dtype=np.uint8)
        for i, x in (rew or {}).items():
            r[i - 1] = x
            v[i - 1] = 1
        rows["reward"].append(r)
        rows["valid"].append(v)
        rows["done"].append(int(done))

    mode_before = {"m": 0}
    try:
        for ep in range(episodes):
            obs = env.reset()
            push(0, None, obs, None, False)
            done = False
            while not done and len(rows["kind"]) < max_rows:
                act = policy(rng, env)
                mode_before["m"] = 0 if env.env.opp_mode == "fight" else 1
                obs, rew, term, trunc, info = env.step(act)
                done = term["__all__"]
                push(1, act, obs, rew, done)
            if len(rows["kind"]) >= max_rows:
                break
    finally:
        _restore_policy_path(cls)
    meta = dict(name=name, env="low", args={k: v for k, v in vars(args).items()}, seed=seed, arena=arena, obs_dim=D,
                draws=len(env.tape.log), policy_files={f: list(NET_FILES[f]) for f in files}, file_names=names, nets_in_loop=True)
    assert len(set(env.tape.log)) == len(env.tape.log), "keyed-RNG key collision"
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"env_{name}.npz")
    np.savez_compressed(path, **out)
    kills = int((np.diff(out["ac_i"][:, :, 0].astype(int), axis=0) < 0).sum())
    dec = np.isfinite(out["opp_margin"])
    print(f"{name}: rows={len(out['kind'])} decisions={int(dec.sum())} min_margin={out['opp_margin'][dec].min():.3g} "
          f"below_1e-5={int((out['opp_margin'][dec] < 1e-5).sum())} files_used={sorted(set(out['opp_file'][dec].tolist()))} "
          f"deaths={kills} distinct_opp_actions={len({tuple(x) for x in out['actions'][:, nA:].reshape(-1, 4).tolist()})} size={os.path.getsize(path)}")


HL_NET_SCENARIOS = [
    # name, args kwargs, files present, episodes, max commander steps
    ("hl_nets_3v3", dict(mode=1), ("L5_AC1_fight.pt", "L5_AC2_fight.pt", "L5_AC1_escape.pt", "L5_AC2_escape.pt"), 3, 60),
    # eval_level_ag = 4 and no L5 escape files: _get_policies falls back to the L3 escape policies (env_base.py:337-343)
    ("hl_nets_2v3", dict(mode=1, num_agents=2, num_opps=3, eval_info=True, horizon=200, eval_level_ag=4),
     ("L4_AC1_fight.pt", "L4_AC2_fight.pt", "L3_AC1_escape.pt", "L3_AC2_escape.pt"), 3, 40),
    # evaluation.py's low-level-vs-low-level mode (eval_hl = False): the opponents fly L{eval_level_opp} fight nets (env_base.py:343-346,387-390)
    ("hl_nets_lowlevel_eval", dict(mode=1, eval_hl=False, eval_level_ag=5, eval_level_opp=4, eval_info=True, horizon=200),
     ("L5_AC1_fight.pt", "L5_AC2_fight.pt", "L5_AC1_escape.pt", "L5_AC2_escape.pt", "L4_AC1_fight.pt", "L4_AC2_fight.pt"), 2, 30),
    # evaluation.py's larger scenarios (README.md:43): five against four on ten unit slots, opponents' target lists of up to five agents (round 5)
    ("hl_nets_5v4", dict(mode=1, num_agents=5, num_opps=4, eval_info=True, horizon=200),
     ("L5_AC1_fight.pt", "L5_AC2_fight.pt", "L5_AC1_escape.pt", "L5_AC2_escape.pt"), 2, 24),
]


def record_hl_nets(name, kw, files, episodes, max_rows, seed=20240917, arena=11):
    """HighLevelEnv with the reference's own pilots in the loop (env_hier.py:114-140 -> env_base.py:349-398): row layout of
    record_hl() plus sub_margin [S, A], sub_logits [S, A, 26] and sub_file [S, A]."""
    import torch
    args = H.make_args(**kw)
    ref = H.load_reference()
    cls = ref["env_hier"].HighLevelEnv
    _restore_policy_path(cls)
    log = []
    real_load = torch.load
    torch.load = reference_policy_loader(set(files), log)
    try:
        env = H.RefEnv("high", args, seed=seed, arena=arena, keep_policies=True)
    finally:
        torch.load = real_load
    A, nA = args.total_num, args.num_agents
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    names = sorted(NET_FILES)
    sub = dict(obs=[], mode=[], act=[], margin=[], logits=[], file=[])
    cur = {"last": 99}
    base_pa = ref["env_base"].HHMARLBaseEnv._policy_actions

    def spy(self_, policy_type, agent_id, unit):
        if agent_id <= cur["last"]:  # first live unit of a new sub-step
            sub["obs"].append(np.zeros((A, 30), dtype=np.float32))
            sub["mode"].append(np.zeros(A, dtype=np.uint8))
            sub["act"].append(np.zeros((A, 4), dtype=np.int8))
            sub["margin"].append(np.full(A, np.inf))
            sub["logits"].append(np.zeros((A, 26), dtype=np.float32))
            sub["file"].append(np.full(A, -1, dtype=np.int8))
        cur["last"] = agent_id
        n0 = len(log)
        out = base_pa(self_, policy_type, agent_id, unit)
        assert len(log) == n0 + 1
        fname, o, logits = log[-1]
        a = np.asarray(out[agent_id])
        sub["obs"][-1][agent_id - 1, : len(o)] = o
        sub["mode"][-1][agent_id - 1] = 1 if policy_type == "fight" else 2
        sub["act"][-1][agent_id - 1, : len(a)] = a
        sub["margin"][-1][agent_id - 1] = decision_margin(logits, len(a))
        sub["logits"][-1][agent_id - 1, : len(logits)] = logits
        sub["file"][-1][agent_id - 1] = names.index(fname)
        return out
    cls._policy_actions = spy
    rows = dict(kind=[], cmd=[], nsub=[], ac_f=[], ac_i=[], rk_f=[], rk_i=[], ar_i=[], tgt_id=[], tgt_d=[], obs=[],
                reward=[], valid=[], done=[], cmd_all=[])
    infos = []

    def push(k, cmd, nsub, obs, rew, done, cmd_all):
        st = env.state()
        rows["kind"].append(k); rows["cmd"].append(cmd); rows["nsub"].append(nsub)
        for key in ("ac_f", "ac_i", "rk_i", "ar_i", "tgt_id", "tgt_d"):
            rows[key].append(st[key])
        rows["rk_f"].append(st["rk_f"][:, :4])
        rows["obs"].append(env.obs_array(obs, 34))
        r = np.zeros(nA); v = np.zeros(nA, dtype=np.uint8)
        for i, x in (rew or {}).items():
            r[i - 1] = x; v[i - 1] = 1
        rows["reward"].append(r); rows["valid"].append(v); rows["done"].append(int(done)); rows["cmd_all"].append(cmd_all)

    try:
        for ep in range(episodes):
            obs = env.reset()
            push(0, np.zeros(nA, dtype=np.int8), 0, obs, None, False, np.zeros(A, dtype=np.int8))
            infos.append({})
            done = False
            while not done and len(rows["kind"]) < max_rows:
                cmd = rng.integers(0, 3, nA).astype(np.int8)
                n0 = len(sub["obs"])
                cur["last"] = 99
                cd = {i + 1: int(cmd[i]) for i in range(nA)}
                obs, rew, term, trunc, info = env.step(cd)
                done = term["__all__"]
                ca = np.array([(cd.get(i) or 0) for i in range(1, A + 1)], dtype=np.int8)
                push(1, cmd, len(sub["obs"]) - n0, obs, rew, done, ca)
                infos.append({k: int(v) for k, v in info.items()})
            if len(rows["kind"]) >= max_rows:
                break
    finally:
        _restore_policy_path(cls)
    meta = dict(name=name, env="high", args={k: v for k, v in vars(args).items()}, seed=seed, arena=arena, obs_dim=34,
                draws=len(env.tape.log), policy_files={f: list(NET_FILES[f]) for f in files}, file_names=names, nets_in_loop=True)
    assert len(set(env.tape.log)) == len(env.tape.log), "keyed-RNG key collision"
    out = {k: np.asarray(v) for k, v in rows.items()}
    for k in sub:
        out["sub_" + k] = np.asarray(sub[k])
    out["infos"] = np.array(json.dumps(infos))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, f"env_{name}.npz")
    np.savez_compressed(path, **out)
    deaths = int((np.diff(out["ac_i"][:, :, 0].astype(int), axis=0) < 0).sum())
    dec = np.isfinite(out["sub_margin"])
    print(f"{name}: rows={len(out['kind'])} substeps={len(sub['obs'])} decisions={int(dec.sum())} min_margin={out['sub_margin'][dec].min():.3g} "
          f"below_1e-5={int((out['sub_margin'][dec] < 1e-5).sum())} files_used={[names[i] for i in sorted(set(out['sub_file'][dec].tolist()))]} "
          f"deaths={deaths} size={os.path.getsize(path)}")


# ---------------------------------------------------------------- hand-built edge cases on the REAL reference
def _geo():
    import geodesic_ref
    return geodesic_ref


def lon_at_range(lat, lon, metres):
    """longitude east of (lat, lon) on the same parallel whose geodesic range from it is `metres` (bisection on the
    reference's own Inverse stand-in; the answer is good to the last few ulps of longitude, ~1e-9 m)"""
    G = _geo()
    lo, hi = lon, lon + 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if G.inverse(lat, lon, lat, mid)[0] < metres:
            lo = mid
        else:
            hi = mid
    return hi


def find_arena(seed, need, start=0):
    """first arena id whose keyed draws (episode 1) satisfy every (tick, unit, site, sub, predicate) in `need`"""
    for arena in range(start, start + 200000):
        ak = H.arena_key(seed, arena)
        if all(pred(H.u01(H.tick_key(ak, 1, t), unit, H.SITES[site], sub)) for t, unit, site, sub, pred in need):
            return arena
    raise RuntimeError("no arena found")


FAR = {2: dict(lat=5.28, lon=7.02, hdg=0.0), 4: dict(lat=5.02, lon=7.28, hdg=180.0, spd=0.0)}   # bystanders out of everybody's way


def edge_cases(seed):
    """name, args kw, arena, inject(units, rockets), per-step agent actions.  Opponents are level-1 (static, env_hetero.py:118-123)
    so that the situation stays exactly as built; thresholds are approached to 1e-5 m / 1e-7 deg — far inside anything a random
    trace reaches, far outside the 1e-9 m agreement of two Karney implementations."""
    G = _geo()
    hit1 = lambda u: u < 0.75 / 5.0          # ac1.py:112-113
    hit2 = lambda u: u < 0.9 / 3.0           # ac2.py:99-100
    noop = {1: [6, 0, 0, 0], 2: [6, 0, 0]}
    cases = []
    # 1. mutual cannon kill in one tick: a unit killed earlier in the tick still shoots (cmano_simulator.py:142)
    ar = find_arena(seed, [(1, 1, "CANNON", 3, hit1), (1, 3, "CANNON", 1, hit1)])
    cases.append(("mutual_cannon_kill", dict(level=1), ar,
                  dict(units={1: dict(lat=5.15, lon=7.15, hdg=90.0, burst=5), 3: dict(lat=5.15, lon=7.16, hdg=270.0, spd=0.0, burst=5), **FAR}),
                  [noop, noop]))
    # 2. rocket fuse on the hard-coded "friendly" id (rocket_unit.py:44-52): agent 1 launches past agent 2, 400 m ahead of it
    cases.append(("fuse_on_friendly_agent_source", dict(level=1), 3,
                  dict(units={1: dict(lat=5.15, lon=7.05, hdg=90.0), 2: dict(lat=5.15, lon=7.0536, hdg=90.0),
                              3: dict(lat=5.15, lon=7.13, hdg=270.0, spd=0.0), 4: FAR[4]}),
                  [{1: [6, 0, 0, 1], 2: [6, 0, 0]}, noop, noop]))
    # 3. the same clause for an opponent's rocket: source id 3 -> "friendly" id 2, i.e. it fuses on AGENT 2 on its way to agent 1
    cases.append(("fuse_on_friendly_opp_source", dict(level=1), 4,
                  dict(units={1: dict(lat=5.15, lon=7.05, hdg=90.0), 2: dict(lat=5.1503, lon=7.10, hdg=0.0),
                              3: dict(lat=5.15, lon=7.16, hdg=270.0, spd=0.0, missile_remain=7), 4: FAR[4]},
                       rockets=[dict(source=3, target=1, lat=5.15, lon=7.112, hdg=270.0, life=2)]),
                  [noop, noop, noop]))
    # 4. failed launch (target outside the radar cone) still draws missile_wait, then decrements it (env_base.py:228-236)
    cases.append(("failed_launch_sets_missile_wait", dict(level=1), 5,
                  dict(units={1: dict(lat=5.15, lon=7.10, hdg=270.0), 3: dict(lat=5.15, lon=7.20, hdg=0.0, spd=0.0), **FAR}),
                  [{1: [6, 0, 0, 1], 2: [6, 0, 0]}] * 4))
    # 5. inclusive map boundary (map_limits.py:47-48): a static opponent exactly ON the southern edge stays, one ulp below it goes
    cases.append(("oob_inclusive_boundary", dict(level=1), 6,
                  dict(units={1: dict(lat=5.15, lon=7.10, hdg=0.0), 2: FAR[2], 3: dict(lat=5.0, lon=7.15, hdg=0.0, spd=0.0),
                              4: dict(lat=float(np.nextafter(5.0, 0.0)), lon=7.25, hdg=0.0, spd=0.0)}),
                  [noop, noop]))
    # 6./7. cannon range 2.0 km (type 1) and 4.5 km (type 2), 1e-5 m inside / outside, with hit draws that would kill
    ar = find_arena(seed, [(1, 1, "CANNON", 3, hit1), (1, 2, "CANNON", 4, hit2)])
    for tag, eps in (("inside", -1e-5), ("outside", +1e-5)):
        cases.append((f"cannon_range_{tag}", dict(level=1), ar,
                      dict(units={1: dict(lat=5.10, lon=7.05, hdg=90.0, burst=5), 3: dict(lat=5.10, lon=lon_at_range(5.10, 7.05, 2000.0 + eps), hdg=0.0, spd=0.0),
                                  2: dict(lat=5.20, lon=7.05, hdg=90.0, burst=3), 4: dict(lat=5.20, lon=lon_at_range(5.20, 7.05, 4500.0 + eps), hdg=0.0, spd=0.0)}),
                      [noop, noop]))
    # 8./9. cannon cone half-width 5 deg (type 1) / 3.5 deg (type 2), 1e-7 deg inside / outside, at 1 km
    for tag, eps in (("inside", -1e-7), ("outside", +1e-7)):
        la3, lo3 = G.direct(5.10, 7.05, 90.0 + 5.0 + eps, 1000.0)
        la4, lo4 = G.direct(5.20, 7.05, 90.0 - 3.5 - eps, 1000.0)
        cases.append((f"cannon_cone_{tag}", dict(level=1), ar,
                      dict(units={1: dict(lat=5.10, lon=7.05, hdg=90.0, burst=5), 3: dict(lat=la3, lon=lo3, hdg=0.0, spd=0.0),
                                  2: dict(lat=5.20, lon=7.05, hdg=90.0, burst=3), 4: dict(lat=la4, lon=lo4, hdg=0.0, spd=0.0)}),
                      [noop, noop]))
    # 10./11. rocket fuse 1.0 km (rocket_unit.py:39), tested BEFORE the rocket moves against the target after ITS move
    for tag, eps in (("inside", -1e-5), ("outside", +1e-5)):
        cases.append((f"rocket_fuse_{tag}", dict(level=1), 8,
                      dict(units={1: dict(lat=5.12, lon=7.02, hdg=90.0, missile_remain=4), 3: dict(lat=5.10, lon=lon_at_range(5.10, 7.05, 1000.0 + eps), hdg=0.0, spd=0.0), **FAR},
                           rockets=[dict(source=1, target=3, lat=5.10, lon=7.05, hdg=90.0, life=3)]),
                      [noop, noop]))
    # 12./13. missile range 111 km inclusive (ac1.py:75) on a 1.2 deg map; 14.-17. the asymmetric radar cone (h - 1, h + 121) deg
    for tag, eps in (("inside", -1e-4), ("outside", +1e-4)):
        cases.append((f"missile_range_{tag}", dict(level=1, map_size=1.2), 9,
                      dict(units={1: dict(lat=5.60, lon=7.05, hdg=90.0), 3: dict(lat=5.60, lon=lon_at_range(5.60, 7.05, 111000.0 + eps), hdg=0.0, spd=0.0),
                                  2: dict(lat=6.10, lon=7.02, hdg=0.0), 4: dict(lat=5.02, lon=8.15, hdg=180.0, spd=0.0)}),
                      [{1: [6, 0, 0, 1], 2: [6, 0, 0]}, noop]))
    for tag, rel in (("low_inside", -1.0 + 1e-7), ("low_outside", -1.0 - 1e-7), ("high_inside", 121.0 - 1e-7), ("high_outside", 121.0 + 1e-7)):
        la3, lo3 = G.direct(5.15, 7.15, 40.0 + rel, 9000.0)
        cases.append((f"radar_cone_{tag}", dict(level=1), 10,
                      dict(units={1: dict(lat=5.15, lon=7.15, hdg=40.0), 3: dict(lat=la3, lon=lo3, hdg=0.0, spd=0.0), **FAR}),
                      [{1: [6, 0, 0, 1], 2: [6, 0, 0]}, noop]))
    return cases


def record_edge(seed=20240917):
    """tests/golden/edge_cases.npz: every case = a reset row (kind 0), an inject row (kind 2: the world snapshot the replay
    loads through hh_set_state, and the observation the reference's state() gives for it) and its step rows (kind 1)."""
    rows = dict(kind=[], case=[], actions=[], ac_f=[], ac_i=[], rk_f=[], rk_i=[], ar_i=[], tgt_id=[], tgt_d=[], obs=[], reward=[],
                valid=[], done=[])
    metas = []
    for ci, (name, kw, arena, inj, script) in enumerate(edge_cases(seed)):
        args = H.make_args(**kw)
        env = H.RefEnv("low", args, seed=seed, arena=arena)
        A, nA = args.total_num, args.num_agents

        def push(k, act, obs, rew, done):
            st = env.state()
            rows["kind"].append(k); rows["case"].append(ci)
            a = np.zeros((A, 4), dtype=np.int8)
            for i, v in (act or {}).items():
                a[i - 1, : len(v)] = v
            rows["actions"].append(a)
            for key in ("ac_f", "ac_i", "rk_i", "tgt_id", "tgt_d"):
                rows[key].append(st[key])
            rows["ar_i"].append(np.concatenate([st["ar_i"], [env.tape.episode]]).astype(np.int32))
            rows["rk_f"].append(st["rk_f"][:, :4])
            rows["obs"].append(env.obs_array(obs, 26))
            r = np.zeros(nA); v = np.zeros(nA, dtype=np.uint8)
            for i, x in (rew or {}).items():
                r[i - 1] = x; v[i - 1] = 1
            rows["reward"].append(r); rows["valid"].append(v); rows["done"].append(int(done))

        push(0, None, env.reset(), None, False)
        push(2, None, env.inject(**inj), None, False)
        for act in script:
            obs, rew, term, trunc, info = env.step(act)
            push(1, act, obs, rew, term["__all__"])
            if term["__all__"]:   # RLlib resets a finished episode; stepping on is outside the boundary's contract
                break
        assert len(set(env.tape.log)) == len(env.tape.log), "keyed-RNG key collision"
        metas.append(dict(name=name, env="low", args={k: v for k, v in vars(args).items()}, seed=seed, arena=arena, obs_dim=26))
    out = {k: np.asarray(v) for k, v in rows.items()}
    out["meta"] = np.array(json.dumps(metas))
    path = os.path.join(OUT, "edge_cases.npz")
    np.savez_compressed(path, **out)
    alive = out["ac_i"][:, :, 0]
    print(f"edge_cases: cases={len(metas)} rows={len(out['kind'])} size={os.path.getsize(path)}")
    for ci, m in enumerate(metas):
        sel = out["case"] == ci
        print(f"   {m['name']:34s} arena={m['arena']:5d} alive after inject {alive[sel][1].tolist()} -> end {alive[sel][-1].tolist()} "
              f"rockets {out['rk_i'][sel][:, :, 0].sum(axis=1).tolist()} wait {out['ac_i'][sel][:, 0, 7].tolist()}")


# ---------------------------------------------------------------- the reference's own per-tick unit trace
def record_unit_trace(name="l3_fight_random", seed=20240917, arena=7):
    """tests/golden/unit_trace_l3.npz: what the reference's simulator RECORDS for rendering — CmanoSimulator.trace_record_units
    (cmano_simulator.py:82,125-130,147-150,159-162: one (utc_time, position, heading, speed) tuple per recorded unit at reset and
    after every tick while the unit exists; reset_sim / env_base.py:581 record every aircraft; rockets enter through add_unit,
    which records nothing, so `ep*_rockets` stays empty) — dumped
    at the end of each episode of the `l3_fight_random` scenario (same seed, arena and action stream as env_l3_fight_random.npz,
    so a replay of that trace must leave the same trajectory in the device-side ring buffer of hh_trace_enable)."""
    sc = [x for x in SCENARIOS if x[0] == name][0]
    _, kind, kw, policy, episodes, max_rows = sc
    args = H.make_args(**kw)
    env = H.RefEnv(kind, args, seed=seed, arena=arena)
    A = args.total_num
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    eps = []
    rows = 0

    def dump():
        sim = env.env.sim
        t0 = min(t for u in range(1, A + 1) for t, _, _, _ in sim.trace_record_units[u])
        T = max(len(v) for v in sim.trace_record_units.values())
        ac = np.full((T, A, 4), np.nan)
        for u in range(1, A + 1):
            for t, pos, hdg, spd in sim.trace_record_units[u]:
                ac[int((t - t0).total_seconds()), u - 1] = (pos.lat, pos.lon, hdg, spd)
        rk = []   # rocket id, source unit, tick, lat, lon
        for uid, tr in sim.trace_record_units.items():
            if uid > A:
                src = env.env._hh_rocket_src[uid]
                for t, pos, hdg, spd in tr:
                    rk.append((uid - A, src, int((t - t0).total_seconds()), pos.lat, pos.lon))
        return ac, np.asarray(rk, dtype=np.float64).reshape(-1, 5)

    for ep in range(episodes):
        env.reset()
        rows += 1
        env.env._hh_rocket_src = {}
        done = False
        while not done and rows < max_rows:
            act = policy(rng, env)
            obs, rew, term, trunc, info = env.step(act)
            for uid, r in env.env.sim.active_units.items():
                if uid > A:
                    env.env._hh_rocket_src[uid] = r.source.id
            rows += 1
            done = term["__all__"]
        eps.append(dump())
        if rows >= max_rows:
            break
    out = {"meta": np.array(json.dumps(dict(name=name, seed=seed, arena=arena, episodes=len(eps))))}
    for k, (ac, rk) in enumerate(eps):
        out[f"ep{k}_aircraft"] = ac
        out[f"ep{k}_rockets"] = rk
    path = os.path.join(OUT, "unit_trace_l3.npz")
    np.savez_compressed(path, **out)
    print(f"unit_trace_l3: episodes={len(eps)} ticks={[len(a) - 1 for a, _ in eps]} rocket points={[len(r) for _, r in eps]} size={os.path.getsize(path)}")


def generate(out_dir, only=()):
    global OUT
    OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    for sc in SCENARIOS:
        if not only or sc[0] in only:
            record(*sc)
    for sc in HL_SCENARIOS:
        if not only or sc[0] in only:
            record_hl(*sc)
    for sc in MINI_SCENARIOS:
        if not only or sc[0] in only:
            record(*sc[:6], seed=sc[6], arena=sc[7])
    for sc in MINI_HL_SCENARIOS:
        if not only or sc[0] in only:
            record_hl(*sc[:5], seed=sc[5], arena=sc[6])
    for sc in NET_SCENARIOS:
        if not only or sc[0] in only:
            record_nets(*sc)
    for sc in HL_NET_SCENARIOS:
        if not only or sc[0] in only:
            record_hl_nets(*sc)
    if not only or "edge_cases" in only:
        record_edge()
    if not only or "unit_trace" in only:
        record_unit_trace()


def check(committed_dir):
    """regenerate every fixture into a temporary directory and compare it, array by array, with the committed files"""
    import glob
    import tempfile
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        generate(tmp)
        new = sorted(os.path.basename(f) for f in glob.glob(os.path.join(tmp, "*.npz")))
        old = sorted(os.path.basename(f) for f in glob.glob(os.path.join(committed_dir, "env_*.npz")) + glob.glob(os.path.join(committed_dir, "edge_*.npz")) + glob.glob(os.path.join(committed_dir, "unit_trace_*.npz")))
        if new != old:
            bad.append(f"file sets differ: generated {new} vs committed {old}")
        for f in new:
            if f not in old:
                continue
            a, b = np.load(os.path.join(tmp, f)), np.load(os.path.join(committed_dir, f))
            if set(a.files) != set(b.files):
                bad.append(f"{f}: keys differ {sorted(set(a.files) ^ set(b.files))}")
                continue
            for k in a.files:
                same = a[k].shape == b[k].shape and (np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k]))
                if not same:
                    bad.append(f"{f}: array {k} differs")
    return bad


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("names", nargs="*", help="scenario names (default: all)")
    ap.add_argument("--out", default=OUT, help="output directory (default tests/golden)")
    ap.add_argument("--check", action="store_true", help="regenerate into a temp dir and compare with tests/golden; exit 1 on any difference")
    a = ap.parse_args()
    if a.check:
        problems = check(a.out)
        print("\n".join(problems) if problems else "golden fixtures reproduce from the committed generator")
        sys.exit(1 if problems else 0)
    generate(a.out, a.names)
