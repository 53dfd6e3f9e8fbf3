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

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_env_golden.py
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
    ("l5_fight_frozen_opps", "low", dict(level=5), pursuit_actions, 4, 360),
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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for sc in SCENARIOS:
        if only and sc[0] not in only:
            continue
        record(*sc)
    for sc in HL_SCENARIOS:
        if only and sc[0] not in only:
            continue
        record_hl(*sc)
