"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(kind="low"):
    """low: LowLevelEnv traces (env_l*.npz); high: HighLevelEnv traces (env_hl_*.npz)"""
    allf = sorted(glob.glob(os.path.join(GOLDEN, "env_*.npz")))
    hl = [f for f in allf if os.path.basename(f).startswith("env_hl_")]
    return hl if kind == "high" else [f for f in allf if f not in hl]


def frozen_opponent_files():
    """LowLevelEnv traces of levels 4-5 (frozen-policy opponents): the taped ones and the ones recorded with the reference's own
    networks in the loop (env_*_nets.npz)"""
    return [f for f in golden_files() if json.loads(str(np.load(f)["meta"]))["args"]["level"] >= 4]


def nets_in_loop_files(kind="low"):
    """traces recorded with the reference's own _get_policies / _policy_actions running (oracle/gen_env_golden.py: record_nets)"""
    return [f for f in golden_files(kind) if json.loads(str(np.load(f)["meta"])).get("nets_in_loop")]


def draws_opponent_policy(meta):
    """env_hetero.py:55-59: only level 5 in FIGHT mode draws k = randint(3,5) per episode (escape-mode level 5 flies the L5 fight
    policies, opp_mode stays "fight": env_hetero.py:50)"""
    return meta["args"]["level"] == 5 and meta["args"]["agent_mode"] == "fight"


def load_golden(path):
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    return g, meta


def edge_cases():
    """tests/golden/edge_cases.npz (hand-built threshold situations run on the REAL reference, oracle/gen_env_golden.py):
    -> list of (meta, rows) with rows a dict of the case's arrays (kind 0 reset, 2 inject = load through set_state, 1 step)"""
    g = np.load(os.path.join(GOLDEN, "edge_cases.npz"))
    metas = json.loads(str(g["meta"]))
    out = []
    for ci, m in enumerate(metas):
        sel = g["case"] == ci
        out.append((m, {k: g[k][sel] for k in g.files if k not in ("meta",)}))
    return out


def edge_state(rows, r):
    """world snapshot (hh_get_state / hh_set_state layout, one arena) of row r of an edge case"""
    return {k: np.ascontiguousarray(rows[k][r][None]) for k in ("ac_f", "ac_i", "rk_f", "rk_i", "ar_i", "tgt_id", "tgt_d")}


def cfg_kwargs_from_meta(meta, **over):
    a = meta["args"]
    kw = dict(
        n_arenas=1, env_kind=0 if meta["env"] == "low" else 1, level=a["level"],
        agent_mode=0 if a["agent_mode"] == "fight" else 1, n_agents=a["num_agents"], n_opps=a["num_opps"],
        horizon=a["horizon"], friendly_kill=a["friendly_kill"], friendly_punish=a["friendly_punish"],
        esc_dist_rew=a["esc_dist_rew"], hier_action_assess=a["hier_action_assess"],
        hier_opp_fight_ratio=a["hier_opp_fight_ratio"], map_size=a["map_size"], glob_frac=a["glob_frac"],
        rew_scale=a["rew_scale"], seed=meta["seed"], arena_offset=meta["arena"],
        ext_opp_actions=(meta["env"] == "low" and a["level"] >= 4),
        opp_side_selector=(meta["env"] == "high" and not a.get("eval_hl", True)))
    kw.update(over)
    return kw


def random_actions(rng, shape_prefix, n_ctrl):
    """uniform MultiDiscrete([13,9,2,2]) actions, int8 [..., n_ctrl, 4]"""
    a = np.zeros(tuple(shape_prefix) + (n_ctrl, 4), dtype=np.int8)
    a[..., 0] = rng.integers(0, 13, size=a.shape[:-1])
    a[..., 1] = rng.integers(0, 9, size=a.shape[:-1])
    a[..., 2] = rng.integers(0, 2, size=a.shape[:-1])
    a[..., 3] = rng.integers(0, 2, size=a.shape[:-1])
    return a


def pursuit_actions(rng, state, n_agents, n_ctrl):
    """Vectorised 'steer at the nearest opponent and shoot' policy on a host snapshot: makes
    cannon/missile engagements (and therefore kill masks) frequent in parity runs."""
    ac_f, ac_i = state["ac_f"], state["ac_i"]
    N, A = ac_f.shape[:2]
    act = np.zeros((N, n_ctrl, 4), dtype=np.int8)
    for i in range(n_ctrl):
        own = i < n_agents
        lo, hi = (n_agents, A) if own else (0, n_agents)
        dlat = ac_f[:, lo:hi, 0] - ac_f[:, i:i + 1, 0]
        dlon = ac_f[:, lo:hi, 1] - ac_f[:, i:i + 1, 1]
        d = np.hypot(dlat, dlon)
        d = np.where(ac_i[:, lo:hi, 0] > 0, d, 1e9)
        j = d.argmin(axis=1)
        idx = np.arange(N)
        brg = np.degrees(np.arctan2(dlon[idx, j], dlat[idx, j])) % 360
        rel = (brg - ac_f[:, i, 2] + 180) % 360 - 180
        act[:, i, 0] = np.clip(np.round(rel / 15.0) + 6, 0, 12)
        act[:, i, 1] = np.where(d[idx, j] < 0.05, rng.integers(0, 4, N), rng.integers(4, 9, N))
        act[:, i, 2] = 1
        act[:, i, 3] = rng.random(N) < 0.5
    return act


# ---- picklable stand-ins for the reference's exported policy modules (torch.save / torch.load round trip in tests)
def _stub_classes():
    import torch.nn as nn

    class StubSlimFC(nn.Module):      # ray's SlimFC keeps its nn.Linear in self._model[0]
        def __init__(self, i, o):
            super().__init__()
            self._model = nn.Sequential(nn.Linear(i, o))

    class StubPolicyNet(nn.Module):   # parameter names of models/ac_models_hetero.py (actor half + a value branch the kernel ignores)
        def __init__(self, inputs, n_out, att):
            super().__init__()
            (a0, a1, w1), (b0, b1, w2), (c0, c1, w3) = inputs
            self.inp1, self.inp2, self.inp3 = StubSlimFC(a1 - a0, w1), StubSlimFC(b1 - b0, w2), StubSlimFC(c1 - c0, w3)
            if att:
                self.att_act = nn.MultiheadAttention(100, 2, batch_first=True)
            self.shared_layer, self.act_out = StubSlimFC(500, 500), StubSlimFC(500, n_out)
            self.v1, self.val_out = StubSlimFC(60, 500), StubSlimFC(500, 1)
    return StubSlimFC, StubPolicyNet


try:
    StubSlimFC, StubPolicyNet = _stub_classes()
    StubSlimFC.__qualname__, StubPolicyNet.__qualname__ = "StubSlimFC", "StubPolicyNet"
except ImportError:   # torch-less environments never reach the tests that use them
    pass


def stub_reference_module(kind, seed):
    """(module, actor state dict) of one synthetic policy with the reference's parameter names"""
    import torch
    from hhmarl_2d_amd import policy_nets as PN
    net = StubPolicyNet(PN.INPUTS[kind], PN.N_OUT[kind], PN.HAS_ATT[kind])
    sd = PN.random_weights(kind, seed)
    params = dict(net.named_parameters())
    with torch.no_grad():
        for k, v in sd.items():
            params[k].copy_(torch.from_numpy(v))
    return net, sd
