"""Composition parity (VERDICT r2 "missing 1"): the reference's OWN frozen-policy path in the loop.

tests/golden/env_*_nets.npz were recorded (oracle/gen_env_golden.py: record_nets / record_hl_nets) with the reference's
`_get_policies` picking the policy files and its `_policy_actions` running — reference env -> lowlevel_state -> reference
Fight1/Fight2/Esc1/Esc2.forward -> get_torch_action -> _take_base_action (env_base.py:312-398, env_hetero.py:160-172,
env_hier.py:114-140).  Here the drop-in facades get a policy directory with the same (synthetic, seeded) weights under the
reference's file names and must reproduce every opponent / pilot ACTION, then state / observation / reward / done — with the
HIP policy kernel (the tile form hh_k_policy_h and the weights-through-LDS form hh_k_policy_w16) inside the loop.

Near-ties are NOT skipped: every decision's top-2 logit margin was recorded; an arg-max that differs from the reference's is
an error unless that margin is <= 1e-5 (north_star's float tolerance on the logits), in which case it is counted, reported
and the recorded action is put back so that the rest of the trace stays comparable."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import draws_opponent_policy, load_golden, nets_in_loop_files, stub_reference_module

pytestmark = pytest.mark.gpu

MARGIN_TOL = 1e-5    # a flipped arg-max is tolerated only below this top-2 logit gap
LOGIT_TOL = 1e-5     # HIP kernel vs the reference's own fp32 forward
FORMS = [pytest.param("0", id="tile-form"), pytest.param("2", id="weights-through-lds")]   # HH_POLICY_W, read at hh_policy_create


def _write_policy_dir(tmp_path, meta):
    """the files the reference's _get_policies would torch.load, holding the weights the recording used"""
    for name, (kind, seed) in meta["policy_files"].items():
        torch.save(stub_reference_module(kind, seed)[0], os.path.join(str(tmp_path), name))
    return str(tmp_path)


def _args(meta, mode):
    from hhmarl_2d_amd.config import make_args
    a = meta["args"]
    keys = ["level", "agent_mode", "horizon", "map_size", "glob_frac", "rew_scale", "esc_dist_rew", "friendly_kill", "friendly_punish"]
    if mode == 1:
        keys += ["num_agents", "num_opps", "hier_action_assess", "hier_opp_fight_ratio", "eval_info", "eval_hl", "eval_level_ag", "eval_level_opp"]
    return make_args(mode, **{k: a[k] for k in keys})


def _with_arena_offset(module, arena):
    """the facades number arenas from 0; a trace was recorded for one global arena id"""
    from hhmarl_2d_amd import env_hetero
    orig = env_hetero.config_from_args
    module.config_from_args = lambda *a, **k: orig(*a, **{**k, "arena_offset": arena})
    return orig


class Flips:
    def __init__(self):
        self.n = 0
        self.decisions = 0
        self.log = []

    def check(self, got, want, margin, where):
        """got / want: int8 [4] actions of one decision.  Returns True when the recorded action has to be put back (near-tie flip)."""
        self.decisions += 1
        if np.array_equal(got, want):
            return False
        assert margin <= MARGIN_TOL, f"{where}: action {got.tolist()} != reference {want.tolist()} at top-2 logit margin {margin:.3g} (> {MARGIN_TOL})"
        self.n += 1
        self.log.append((where, float(margin)))
        return True


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("path", nets_in_loop_files("low"), ids=lambda p: os.path.basename(p)[4:-4])
def test_lowlevel_env_with_policy_dir_reproduces_the_reference_trace(path, form, tmp_path, monkeypatch):
    """LowLevelEnv(level 4 / 5, policy_dir): opponents' actions from the HIP policy kernel == the reference's own networks'"""
    from hhmarl_2d_amd import env_hetero
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    monkeypatch.setenv("HH_POLICY_W", form)
    g, meta = load_golden(path)
    pdir = _write_policy_dir(tmp_path, meta)
    orig = _with_arena_offset(env_hetero, meta["arena"])
    try:
        env = LowLevelEnv({"args": _args(meta, 0), "seed": meta["seed"], "policy_dir": pdir})
    finally:
        env_hetero.config_from_args = orig
    inner = env.opponent_policy
    flips, cur = Flips(), {"r": 0}

    def checked(opp_obs, e):
        r = cur["r"]
        assert np.abs(opp_obs.cpu().numpy()[0] - g["opp_obs"][r]).max() <= 1e-6, f"row {r}: opponents' policy observation"
        act = inner(opp_obs, e)
        got = act.cpu().numpy()[0]
        for j in range(2):
            if np.isfinite(g["opp_margin"][r][j]):     # this opponent existed and decided
                if flips.check(got[j], g["actions"][r][2 + j], g["opp_margin"][r][j], f"row {r} opponent {3 + j}"):
                    act[0, j] = torch.from_numpy(np.ascontiguousarray(g["actions"][r][2 + j])).to(act.device)
        return act
    env.opponent_policy = checked
    dims = env.obs_dim_map
    for r in range(len(g["kind"])):
        cur["r"] = r
        if g["kind"][r] == 0:
            obs, _ = env.reset()
        else:
            if draws_opponent_policy(meta):
                assert env.opp_mode == ("escape" if g["opp_mode"][r] == 1 else "fight"), f"row {r}: level-5 policy draw"
            obs, rew, term, trunc, info = env.step({1: g["actions"][r][0, :4].tolist(), 2: g["actions"][r][1, :3].tolist()})
            assert term["__all__"] == bool(g["done"][r]), f"row {r}: done"
            assert set(rew) == {i + 1 for i in range(2) if g["valid"][r][i]}, f"row {r}: reward keys"
            for i in rew:
                assert abs(rew[i] - g["reward"][r][i - 1]) <= 1e-6 * max(1.0, abs(g["reward"][r][i - 1])), f"row {r}: reward"
            st = env.world.get_state()
            assert np.array_equal(st["ac_i"][0], g["ac_i"][r]) and np.array_equal(st["rk_i"][0], g["rk_i"][r]), f"row {r}: integer state"
            assert np.abs(st["ac_f"][0] - g["ac_f"][r]).max() <= 1e-9, f"row {r}: aircraft floats"
        for i in (1, 2):
            assert np.abs(obs[i] - g["obs"][r][i - 1, : dims[i]]).max() <= 1e-6, f"row {r}: observation"
    n_dec = int(np.isfinite(g["opp_margin"]).sum())
    assert flips.decisions == n_dec and n_dec > 300
    print(f"{os.path.basename(path)} [{'hh_k_policy_w16' if form == '2' else 'hh_k_policy_h'}]: {n_dec} decisions reproduced, {flips.n} near-tie flips {flips.log}")
    assert flips.n <= 2
    env.close()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("rows", ["variants", "sides"])
@pytest.mark.parametrize("path", nets_in_loop_files("high"), ids=lambda p: os.path.basename(p)[4:-4])
def test_highlevel_env_with_policy_dir_reproduces_the_reference_trace(path, form, rows, tmp_path, monkeypatch):
    """HighLevelEnv(policy_dir): every pilot's action of every sub-step from the HIP policy kernel == the reference's networks';
    commander observations, rewards, done and eval_info follow.  Covers the L5 -> L3 escape fallback (hl_nets_2v3) and
    evaluation.py's eval_hl = False mode, where the opponents fly their own L{eval_level_opp} fight nets (hl_nets_lowlevel_eval)."""
    import hhmarl_2d_amd.env_hier as eh
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    monkeypatch.setenv("HH_POLICY_W", form)
    g, meta = load_golden(path)
    infos = json.loads(str(g["infos"]))
    a = meta["args"]
    nA, A = a["num_agents"], a["num_agents"] + a["num_opps"]
    pdir = _write_policy_dir(tmp_path, meta)
    orig = _with_arena_offset(eh, meta["arena"])
    try:
        env = HighLevelEnv({"args": _args(meta, 1), "seed": meta["seed"], "policy_dir": pdir, "pilot_rows": rows})
    finally:
        eh.config_from_args = orig
    inner = env.pilot
    variants = bool(getattr(inner, "variants", False))   # (ten-slot worlds fly the two-call form whatever was asked)
    flips, tape = Flips(), {"calls": 0}
    files = meta["file_names"]
    want_files = set()

    def checked_variants(po, pm):
        """the variant-row form (the facade's default): ONE pilot call per sub-step over [1, 15, 30] — the agents' rows are the recorded ones; each
        opponent's recorded row (what the reference's pilot saw AFTER the agents acted) is one of its listed variants, and that variant's action is the
        recorded one"""
        k = tape["calls"]
        tape["calls"] += 1
        assert k < len(g["sub_act"]), "more pilot calls than the reference made sub-steps"
        pm_h, po_h = pm.cpu().numpy()[0], po.cpu().numpy()[0]
        assert np.array_equal(pm_h[:nA] & 3, g["sub_mode"][k][:nA]), f"sub-step {k}: agents' policy types"
        assert np.abs(po_h[:nA] - g["sub_obs"][k][:nA]).max() <= 1e-6, f"sub-step {k}: agents' pilot observations"
        act = inner(po, pm)
        got = act.cpu().numpy()[0]
        for j in range(A):
            if not g["sub_mode"][k][j]:
                continue
            want_files.add(files[g["sub_file"][k][j]])
            if j < nA:
                slots = [j]
            else:   # the variant that matches what the reference's pilot observed
                base = 3 + 4 * (j - nA)
                slots = [base + v for v in range(4) if pm_h[base + v] and np.abs(po_h[base + v] - g["sub_obs"][k][j]).max() <= 1e-6]
                assert slots, f"sub-step {k} unit {j + 1}: the recorded observation is not among the listed variants"
                assert all((pm_h[q] & 3) == g["sub_mode"][k][j] for q in slots)
                if a["eval_hl"] is False:
                    assert all(((pm_h[q] & 64) != 0) == (g["sub_mode"][k][j] == 1) for q in slots), "side bit on the opponents' fight rows"
                slots = slots[:1]
            if flips.check(got[slots[0]], g["sub_act"][k][j], g["sub_margin"][k][j], f"sub-step {k} unit {j + 1}"):
                fix = torch.from_numpy(np.ascontiguousarray(g["sub_act"][k][j])).to(act.device)
                if j < nA:
                    act[0, j] = fix
                else:
                    act[0, 3 + 4 * (j - nA): 7 + 4 * (j - nA)] = fix
        return act
    checked_variants.variants = True

    def checked(po, pm):
        k, side = tape["calls"] // 2, tape["calls"] % 2      # sub-step record, 0 = agents' call, 1 = opponents' call
        tape["calls"] += 1
        assert k < len(g["sub_act"]), "more pilot calls than the reference made sub-steps"
        lo, hi = (0, nA) if side == 0 else (nA, A)
        pm_h, po_h = pm.cpu().numpy()[0], po.cpu().numpy()[0]
        assert np.array_equal(pm_h[lo:hi] & 3, g["sub_mode"][k][lo:hi]), f"sub-step {k} side {side}: policy types"
        assert np.abs(po_h[lo:hi] - g["sub_obs"][k][lo:hi]).max() <= 1e-6, f"sub-step {k} side {side}: pilot observations"
        if a["eval_hl"] is False and side == 1:
            assert ((pm_h[lo:hi] & 64) != 0).tolist() == (g["sub_mode"][k][lo:hi] == 1).tolist(), "side bit on the opponents' fight rows"
        act = inner(po, pm)
        got = act.cpu().numpy()[0]
        for j in range(lo, hi):
            if g["sub_mode"][k][j]:
                want_files.add(files[g["sub_file"][k][j]])
                if flips.check(got[j], g["sub_act"][k][j], g["sub_margin"][k][j], f"sub-step {k} unit {j + 1}"):
                    act[0, j] = torch.from_numpy(np.ascontiguousarray(g["sub_act"][k][j])).to(act.device)
        return act
    env.pilot = checked_variants if variants else checked
    for r in range(len(g["kind"])):
        if g["kind"][r] == 0:
            obs, info = env.reset()
        else:
            obs, rew, term, trunc, info = env.step({i + 1: int(g["cmd"][r][i]) for i in range(nA)})
            assert term["__all__"] == bool(g["done"][r]), f"row {r}: done"
            if a["eval_info"]:
                assert info == infos[r], f"row {r}: eval info {info} != {infos[r]}"
            for i in rew:
                assert abs(rew[i] - g["reward"][r][i - 1]) <= 1e-6, f"row {r}: reward"
            st = env.world.get_state()
            assert np.array_equal(st["rk_i"][0][:A], g["rk_i"][r]) and np.array_equal(st["ac_i"][0][:A, :9], g["ac_i"][r][:, :9]), f"row {r}: integer state"
            assert np.abs(st["ac_f"][0][:A] - g["ac_f"][r]).max() <= 1e-9, f"row {r}: aircraft floats"
        for i in range(1, nA + 1):
            assert np.abs(obs[i] - g["obs"][r][i - 1]).max() <= 1e-6, f"row {r}: commander observation"
    assert tape["calls"] == (1 if variants else 2) * len(g["sub_act"]), "sub-step count"
    n_dec = int((g["sub_mode"] != 0).sum())
    assert flips.decisions == n_dec and n_dec > 1500
    assert want_files == set(meta["policy_files"]), "every loaded policy file flew at least once"
    print(f"{os.path.basename(path)} [{'hh_k_policy_w16' if form == '2' else 'hh_k_policy_h'}, {'variant rows' if variants else 'two calls per sub-step'}]: {n_dec} decisions reproduced, {flips.n} near-tie flips {flips.log}")
    assert flips.n <= 3
    env.close()


@pytest.mark.parametrize("form", FORMS)
def test_policy_kernel_logits_equal_the_reference_networks_on_recorded_observations(form, monkeypatch):
    """every (observation row, logits) pair the reference's networks produced INSIDE the recorded environments, through the HIP
    kernel in one batch per trace: logits <= 1e-5, arg-max equal wherever the top-2 margin exceeds 1e-5 — no row is skipped"""
    from hhmarl_2d_amd import policy_nets as PN
    from hhmarl_2d_amd.pilots import PolicyBank
    monkeypatch.setenv("HH_POLICY_W", form)
    total = worst = 0
    for kind_ in ("low", "high"):
        for path in nets_in_loop_files(kind_):
            g, meta = load_golden(path)
            names = meta["file_names"]
            if kind_ == "low":
                obs, lg, fl, mg = g["opp_obs"].reshape(-1, 30), g["opp_logits"].reshape(-1, 26), g["opp_file"].reshape(-1), g["opp_margin"].reshape(-1)
                act = g["actions"][:, 2:].reshape(-1, 4)
            else:
                obs, lg, fl, mg = g["sub_obs"].reshape(-1, 30), g["sub_logits"].reshape(-1, 26), g["sub_file"].reshape(-1), g["sub_margin"].reshape(-1)
                act = g["sub_act"].reshape(-1, 4)
            keep = fl >= 0
            obs, lg, fl, mg, act = obs[keep], lg[keep], fl[keep], mg[keep], act[keep]
            used = sorted(set(fl.tolist()))
            bank = PolicyBank(torch.device("cuda", 0), max_rows=len(obs))
            lut = {}
            for slot, f in enumerate(used):
                kind, seed = meta["policy_files"][names[f]]
                bank.set_net(slot, kind, PN.random_weights(kind, seed))
                lut[100 + f] = slot
            bank.set_lut(lut)
            sel = torch.from_numpy((100 + fl).astype(np.uint8)).cuda()
            out = torch.zeros((len(obs), 32), device="cuda")
            got = bank.act(torch.from_numpy(np.ascontiguousarray(obs)).cuda(), sel, logits=out).cpu().numpy()
            out = out.cpu().numpy()
            for f in used:
                kind, _ = meta["policy_files"][names[f]]
                rows = fl == f
                err = np.abs(out[rows, : PN.N_OUT[kind]] - lg[rows, : PN.N_OUT[kind]]).max()
                worst = max(worst, float(err))
                assert err <= LOGIT_TOL, f"{os.path.basename(path)} {names[f]}: logits differ from the reference's by {err:.3g}"
            bad = (got != act).any(axis=1)
            assert (mg[bad] <= MARGIN_TOL).all(), f"{os.path.basename(path)}: arg-max differs at margins {mg[bad]}"
            total += len(obs)
            bank.close()
    assert total > 8000
    print(f"{total} recorded decisions, worst logit error {worst:.3g}")
