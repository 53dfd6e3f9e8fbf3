"""The frozen pilot / opponent networks.  CPU part: the plain-PyTorch fp32 restatement of the actor forward
(hhmarl_2d_amd/policy_nets.py) against vectors recorded from the REAL reference model classes called the way the
environment calls them (oracle/gen_policy_golden.py -> tests/golden/policy_nets.npz): logits <= 1e-5, actions exact.
GPU part (-m gpu): the fused HIP kernel against the same vectors and against the PyTorch restatement at larger sizes."""
import os

import numpy as np
import pytest
import torch

from helpers import random_actions

from hhmarl_2d_amd import policy_nets as PN
import policy_ref as PR   # oracle/policy_ref.py: the fp32 PyTorch restatement (test infrastructure)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_nets.npz")
LOGIT_TOL = 1e-5


@pytest.mark.parametrize("kind", [PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2], ids=lambda k: PN.KIND_NAMES[k])
def test_torch_restatement_matches_reference_classes(kind):
    g = np.load(GOLD)
    name = PN.KIND_NAMES[kind].lower()
    sd = PN.random_weights(kind, int(g["seed"]))
    obs = torch.from_numpy(g[f"obs_{name}"])
    logits = PR.torch_forward(kind, sd, obs)
    assert logits.shape == (obs.shape[0], PN.N_OUT[kind])
    assert np.abs(logits.numpy() - g[f"logits_{name}"]).max() <= LOGIT_TOL
    assert np.array_equal(PR.decode(logits, PN.N_OUT[kind]).numpy(), g[f"act_{name}"])


def test_weight_tables_are_consistent():
    for kind in (PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2):
        (a0, a1, w1), (b0, b1, w2), (c0, c1, w3) = PN.INPUTS[kind]
        assert w1 + w2 + w3 == 500 and max(a1, b1, c1) == PN.OBS_DIM[kind]
        sd = PN.random_weights(kind, 1)
        assert set(sd) == set(PN.actor_keys(kind)) and all(v.dtype == np.float32 for v in sd.values())
        assert np.array_equal(sd["act_out._model.0.bias"], PN.random_weights(kind, 1)["act_out._model.0.bias"])
        assert PN.flops_per_row(kind) > 5e5


# ---------------------------------------------------------------------------------------------- GPU: the fused HIP kernel
def _bank(seed, max_rows=1 << 16):
    from hhmarl_2d_amd.pilots import PolicyBank
    return PolicyBank.random_init(torch.device("cuda", 0), seed=seed, max_rows=max_rows)


FORMS = {"split-fp16": {}, "split-fp16-64-row-tiles": {"HH_POLICY_TILE": "64"},
         "weights-through-lds-16-rows": {"HH_POLICY_W": "2"}, "weights-through-lds-16-rows-8-waves": {"HH_POLICY_W": "3"}}


def _form(monkeypatch, form):
    """the kernel form a bank created from now on runs (read at hh_policy_create)"""
    for k in ("HH_POLICY_TILE", "HH_POLICY_W"):
        monkeypatch.delenv(k, raising=False)
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)


@pytest.mark.gpu
@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("kind", [PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2], ids=lambda k: PN.KIND_NAMES[k])
def test_hip_kernel_matches_reference_vectors(monkeypatch, kind, form):
    """hh_policy_act (every form of the forward kernel) against the vectors recorded from the reference's own model classes: logits
    1e-5, actions exact"""
    from hhmarl_2d_amd import pilots
    _form(monkeypatch, form)
    g = np.load(GOLD)
    name = PN.KIND_NAMES[kind].lower()
    bank = _bank(int(g["seed"]))
    obs = torch.zeros((96, 30), dtype=torch.float32)
    obs[:, : PN.OBS_DIM[kind]] = torch.from_numpy(g[f"obs_{name}"])
    sel_byte = {PN.FIGHT1: pilots.SEL_FIGHT1, PN.FIGHT2: pilots.SEL_FIGHT2, PN.ESC1: pilots.SEL_ESC1, PN.ESC2: pilots.SEL_ESC2}[kind]
    sel = torch.full((96,), sel_byte, dtype=torch.uint8, device="cuda")
    logits = torch.full((96, 32), 7.0, dtype=torch.float32, device="cuda")
    act = bank.act(obs.cuda(), sel, logits=logits).cpu().numpy()
    lg = logits.cpu().numpy()
    assert np.abs(lg[:, : PN.N_OUT[kind]] - g[f"logits_{name}"]).max() <= LOGIT_TOL
    assert (lg[:, PN.N_OUT[kind]:] == 0).all()
    assert np.array_equal(act, g[f"act_{name}"])


@pytest.mark.gpu
def test_hip_kernel_mixed_networks_against_torch_fp32():
    """all four networks interleaved row by row, rows without a network, a row count that is no multiple of the tile, strided
    observations: every row against the plain PyTorch fp32 forward of its network"""
    from hhmarl_2d_amd import pilots
    R, D = 20011, 30
    bank = _bank(3, max_rows=R)
    rng = np.random.default_rng(0)
    obs = torch.from_numpy(rng.random((R, D)).astype(np.float32)).cuda()
    sels = np.array([0, pilots.SEL_FIGHT1, pilots.SEL_FIGHT2, pilots.SEL_ESC1, pilots.SEL_ESC2, 77], dtype=np.uint8)
    sel = torch.from_numpy(sels[rng.integers(0, len(sels), R)]).cuda()
    logits = torch.full((R, 32), -3.0, dtype=torch.float32, device="cuda")
    act = bank.act(obs, sel, logits=logits)
    torch.cuda.synchronize()
    kinds = {pilots.SEL_FIGHT1: PN.FIGHT1, pilots.SEL_FIGHT2: PN.FIGHT2, pilots.SEL_ESC1: PN.ESC1, pilots.SEL_ESC2: PN.ESC2}
    checked = 0
    for byte, kind in kinds.items():
        idx = (sel == byte).nonzero().flatten()
        x = torch.zeros((len(idx), D), device="cuda")
        x[:, : PN.OBS_DIM[kind]] = obs[idx, : PN.OBS_DIM[kind]]     # the kernel must ignore columns beyond the net's width
        ref = PR.torch_forward(kind, PN.random_weights(kind, 3), x.cpu())          # CPU fp32, like the reference
        got = logits[idx, : PN.N_OUT[kind]].cpu()
        assert (got - ref).abs().max() <= LOGIT_TOL, PN.KIND_NAMES[kind]
        ra = PR.decode(ref, PN.N_OUT[kind])
        # arg-max must agree wherever the reference's winner leads by more than the tolerance
        parts = ref.split(PN.ACTION_SPLIT[: 4 if PN.N_OUT[kind] == 26 else 3], dim=1)
        clear = torch.stack([(p.topk(2, dim=1).values[:, 0] - p.topk(2, dim=1).values[:, 1]) > 1e-4 for p in parts], dim=1).all(dim=1)
        assert torch.equal(act[idx].cpu()[clear], ra[clear]) and clear.float().mean() > 0.99
        checked += len(idx)
    none = ((sel == 0) | (sel == 77)).nonzero().flatten()
    assert (act[none] == 0).all() and (logits[none] == -3.0).all()
    assert checked + len(none) == R
    # a second call on the same bank (counters are re-zeroed on the stream) gives the same answer, and so does a call that
    # re-uses the row lists (sel = None: "same selectors as before") on changed observations
    assert torch.equal(bank.act(obs, sel), act)
    obs2 = obs.flip(0).contiguous()
    want = bank.act(obs2, sel).clone()
    bank.act(obs, sel)
    got = bank.act(obs2, None)
    live = (sel != 0) & (sel != 77)
    assert torch.equal(got[live], want[live])


@pytest.mark.gpu
@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("mode", ["fight", "escape"])
def test_every_form_on_real_observations_of_a_full_world(monkeypatch, form, mode):
    """16384 arenas x 2 agents a few ticks into their episodes — real observation rows (exact 0 / 1 flags, clipped values, zero friend
    blocks), not uniform noise: every form of the forward kernel against the plain PyTorch fp32 forward, logits 1e-5 on EVERY row.  (Uniform
    random rows let a folded reciprocal square root in the attention block's normalisation pass: 0.2 % of real rows were 2e-5 off.)"""
    from hhmarl_2d_amd import _lib as L, pilots
    from hhmarl_2d_amd.world import World, make_config
    _form(monkeypatch, form)
    N = 16384
    w = World(make_config(n_arenas=N, level=3, agent_mode=L.MODE_FIGHT if mode == "fight" else L.MODE_ESCAPE, seed=77, arena_offset=1000, auto_reset=True), device=0)
    obs = w.reset()
    rng = np.random.default_rng(1)
    for _ in range(3):
        a = torch.from_numpy(np.stack([rng.integers(0, 13, (N, 2)), rng.integers(0, 9, (N, 2)), rng.integers(0, 2, (N, 2)), rng.integers(0, 2, (N, 2))],
                                      axis=-1).astype(np.int8)).cuda()
        obs = w.step(a)[0]
    bank = _bank(5, max_rows=2 * N)
    kinds = (PN.FIGHT1, PN.FIGHT2) if mode == "fight" else (PN.ESC1, PN.ESC2)
    bytes_ = (pilots.SEL_FIGHT1, pilots.SEL_FIGHT2) if mode == "fight" else (pilots.SEL_ESC1, pilots.SEL_ESC2)
    sel = torch.tensor(bytes_, dtype=torch.uint8, device="cuda").repeat(N, 1).contiguous()
    logits = torch.zeros((N, 2, 32), dtype=torch.float32, device="cuda")
    act = bank.act(obs, sel, logits=logits)
    torch.cuda.synchronize()
    o = obs.cpu()
    for slot, kind in enumerate(kinds):
        ref = PR.torch_forward(kind, PN.random_weights(kind, 5), o[:, slot])
        err = (logits[:, slot, : PN.N_OUT[kind]].cpu() - ref).abs()
        assert err.max() <= LOGIT_TOL, f"{PN.KIND_NAMES[kind]}: {float(err.max()):.2e} on {int((err.max(dim=1).values > LOGIT_TOL).sum())} rows"
        parts = ref.split(PN.ACTION_SPLIT[: 4 if PN.N_OUT[kind] == 26 else 3], dim=1)
        clear = torch.stack([(p.topk(2, dim=1).values[:, 0] - p.topk(2, dim=1).values[:, 1]) > 1e-4 for p in parts], dim=1).all(dim=1)
        assert torch.equal(act[:, slot].cpu()[clear], PR.decode(ref, PN.N_OUT[kind])[clear])


@pytest.mark.gpu
def test_net_pilot_drives_highlevel_env_and_matches_torch():
    """NetPilot inside the HighLevelEnv macro step: the selector bytes the world emits (policy type | aircraft type << 2) pick the
    network; actions equal the PyTorch forward of the same rows"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.world import World, make_config
    w = World(make_config(n_arenas=500, env_kind=1, seed=5, auto_reset=True))
    w.reset()
    pilot = pilots.NetPilot(w, seed=9)
    cmd = torch.from_numpy(np.random.default_rng(1).integers(0, 3, (500, 3)).astype(np.int8)).cuda()
    po, pm = w.hl_begin(cmd)
    act = pilot(po, pm).clone()
    kinds = {pilots.SEL_FIGHT1: PN.FIGHT1, pilots.SEL_FIGHT2: PN.FIGHT2, pilots.SEL_ESC1: PN.ESC1, pilots.SEL_ESC2: PN.ESC2}
    seen = 0
    for byte, kind in kinds.items():
        idx = (pm == byte).nonzero()
        if len(idx) == 0:
            continue
        rows = po[idx[:, 0], idx[:, 1]].cpu()
        ref = PR.torch_forward(kind, PN.random_weights(kind, 9), rows)
        parts = ref.split(PN.ACTION_SPLIT[: 4 if PN.N_OUT[kind] == 26 else 3], dim=1)
        clear = torch.stack([(p.topk(2, dim=1).values[:, 0] - p.topk(2, dim=1).values[:, 1]) > 1e-4 for p in parts], dim=1).all(dim=1)
        assert torch.equal(act[idx[:, 0], idx[:, 1]].cpu()[clear], PR.decode(ref, PN.N_OUT[kind])[clear])
        seen += 1
    assert seen >= 3 and (act[pm == 0] == 0).all()
    from hhmarl_2d_amd.env_hier import macro_step
    for _ in range(3):   # whole macro steps run with the networks in the loop
        obs, rew, val, done = macro_step(w, cmd, pilot)
    assert torch.isfinite(obs).all()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [4, 5])
def test_opponent_nets_drive_levels_4_5_through_the_facade(level):
    """LowLevelEnv levels 4-5 with the frozen opponent policies in the fused kernel (env_base.py:312-398, env_hetero.py:160-172):
    the facade hands the opponents' observations to OpponentNets, whose actions equal the PyTorch forward of the network the
    arena's level-5 draw selects (fight nets, or the escape nets when k == 5)"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    n = 256
    seen = {"calls": 0, "esc": 0}
    holder = {}

    def policy(opp_obs, env):
        if "nets" not in holder:   # created inside the first callback, i.e. after the first hh_step_begin already ran
            holder["nets"] = pilots.OpponentNets(env.world, seed=11)
        nets = holder["nets"]
        act = nets(opp_obs, env).clone()
        k = env.world.opp_policy().cpu().numpy()
        for slot, (fight, esc) in enumerate(((PN.FIGHT1, PN.ESC1), (PN.FIGHT2, PN.ESC2))):
            for kind, sel_rows in ((fight, k != 5), (esc, k == 5)):
                idx = np.nonzero(sel_rows)[0]
                if len(idx) == 0:
                    continue
                idx = idx[(opp_obs[idx, slot].abs().sum(dim=1) > 0).cpu().numpy()]   # live opponents of running arenas (others get no action)
                if len(idx) == 0:
                    continue
                ref = PR.torch_forward(kind, PN.random_weights(kind, 11), opp_obs[idx, slot].cpu())
                parts = ref.split(PN.ACTION_SPLIT[: 4 if PN.N_OUT[kind] == 26 else 3], dim=1)
                clear = torch.stack([(p.topk(2, dim=1).values[:, 0] - p.topk(2, dim=1).values[:, 1]) > 1e-4 for p in parts], dim=1).all(dim=1)
                assert torch.equal(act[idx, slot].cpu()[clear], PR.decode(ref, PN.N_OUT[kind])[clear])
                seen["esc"] += int(kind in (PN.ESC1, PN.ESC2)) * len(idx)
        seen["calls"] += 1
        return act

    env = LowLevelEnv({"args": make_args(0, level=level, horizon=30), "num_envs": n, "seed": 3, "opponent_policy": policy})
    obs, _ = env.reset()
    rng = np.random.default_rng(0)
    for t in range(12):
        obs, rew, term, _, _ = env.step({1: np.stack([rng.integers(0, 13, n), rng.integers(0, 9, n), rng.integers(0, 2, n), rng.integers(0, 2, n)], axis=1),
                                         2: np.stack([rng.integers(0, 13, n), rng.integers(0, 9, n), rng.integers(0, 2, n)], axis=1)})
    assert seen["calls"] == 12 and obs[1].shape == (n, 26)
    if level == 5:
        assert seen["esc"] > 0 and set(np.unique(env.opp_k)) <= {3, 4, 5}   # roughly a third of the arenas drew the escape set
    else:
        assert seen["esc"] == 0
    env.close()


@pytest.mark.gpu
def test_policy_abi_misuse_is_reported_through_status_codes():
    import ctypes as C
    from hhmarl_2d_amd import _lib as L
    lib = L.lib()
    h = C.c_void_p()
    assert lib.hh_policy_create(0, 0, C.byref(h)) < 0 and lib.hh_last_error()            # max_rows <= 0
    assert lib.hh_policy_create(99, 64, C.byref(h)) < 0                                   # no such device
    assert lib.hh_policy_create(0, 64, C.byref(h)) == 0
    obs = torch.zeros((64, 30), device="cuda"); sel = torch.zeros((64,), dtype=torch.uint8, device="cuda"); act = torch.zeros((64, 4), dtype=torch.int8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.hh_policy_act(h, p(obs), 64, 30, p(sel), p(act), None, None) < 0          # no network loaded
    lut = np.zeros(256, dtype=np.uint8); lut[5] = 1
    assert lib.hh_policy_set_lut(h, lut.ctypes.data_as(C.c_void_p)) < 0                   # selector maps to an empty slot
    w = L.HHNetWeights(); w.kind = 7
    assert lib.hh_policy_set_net(h, 0, C.byref(w)) < 0                                    # bad kind / missing pointers
    assert lib.hh_policy_destroy(h) == 0
    from hhmarl_2d_amd.pilots import PolicyBank
    bank = PolicyBank.random_init(torch.device("cuda", 0), seed=1, max_rows=64)
    assert lib.hh_policy_act(bank.h, p(obs), 65, 30, p(sel), p(act), None, None) < 0      # more rows than max_rows
    assert lib.hh_policy_act(bank.h, p(obs), 64, 30, None, p(act), None, None) < 0        # sel == NULL before any binning call
    bank.act(obs, sel)                                                                      # the bank is still usable


@pytest.mark.gpu
def test_four_and_eight_wave_instances_of_the_streamed_form_agree_bit_for_bit(monkeypatch):
    """hh_k_policy_w16<4> (64-row tiles, two workgroups per CU, one chunk ahead) and hh_k_policy_w16<8> (128-row tiles, a ring of four chunk buffers three
    chunks ahead, the hand-over inside the chunk before) run the same per-wave arithmetic in the same order: identical logits and actions on a mixed batch
    with ragged last tiles; and the row count picks <8> exactly when 128-row tiles fill whole rounds of the CUs"""
    from hhmarl_2d_amd import pilots
    rng = np.random.default_rng(11)
    sels = np.array([0, pilots.SEL_FIGHT1, pilots.SEL_FIGHT2, pilots.SEL_ESC1, pilots.SEL_ESC2], dtype=np.uint8)
    R = 20011
    obs = torch.from_numpy(rng.random((R, 30)).astype(np.float32)).cuda()
    sel = torch.from_numpy(sels[rng.integers(0, len(sels), R)]).cuda()
    res = []
    for w in ("2", "3"):
        monkeypatch.setenv("HH_POLICY_W", w)
        bank = _bank(7, max_rows=R)
        assert bank.kernel_name(R) == ("hh_k_policy_w16<4>" if w == "2" else "hh_k_policy_w16<8>")
        lg = torch.zeros((R, 32), device="cuda")
        act = bank.act(obs, sel, logits=lg).clone()
        torch.cuda.synchronize()
        res.append((lg, act))
        bank.close()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    monkeypatch.delenv("HH_POLICY_W", raising=False)
    bank = _bank(7, max_rows=65536)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert bank.kernel_name(128 * n_cu) == "hh_k_policy_w16<8>" and bank.kernel_name(128 * n_cu + 64 * n_cu) == "hh_k_policy_w16<4>"
    assert bank.kernel_name(64 * n_cu) == "hh_k_policy_w16<4>" and bank.kernel_name(32 * n_cu) in ("hh_k_policy_h<1>", "hh_k_policy_h<2>")
    bank.close()


@pytest.mark.gpu
def test_tile_instances_of_the_kernel_agree_bit_for_bit(monkeypatch):
    """the 64-row-tile persistent instance (HH_POLICY_TILE=64, with and without the grid-stride walk) computes every row with the
    same operation order as the default 32-row instance: identical logits and actions on a mixed batch larger than one round of
    workgroups, and on a ragged small one"""
    from hhmarl_2d_amd import pilots
    rng = np.random.default_rng(5)
    sels = np.array([0, pilots.SEL_FIGHT1, pilots.SEL_FIGHT2, pilots.SEL_ESC1, pilots.SEL_ESC2], dtype=np.uint8)
    for R in (40003, 77):
        obs = torch.from_numpy(rng.random((R, 30)).astype(np.float32)).cuda()
        sel = torch.from_numpy(sels[rng.integers(0, len(sels), R)]).cuda()
        res = []
        for tile, persist in (("32", "1"), ("64", "1"), ("64", "0")):
            monkeypatch.setenv("HH_POLICY_TILE", tile)
            monkeypatch.setenv("HH_POLICY_PERSIST", persist)
            bank = _bank(7, max_rows=R)
            lg = torch.zeros((R, 32), device="cuda")
            act = bank.act(obs, sel, logits=lg).clone()
            torch.cuda.synchronize()
            res.append((lg, act))
            bank.close()
        for lg, act in res[1:]:
            assert torch.equal(lg, res[0][0]) and torch.equal(act, res[0][1])
    # the width can also be pinned per bank (hh_policy_set_tile_rows) or left to the row count (0): same bits again, bad values refused
    monkeypatch.delenv("HH_POLICY_TILE", raising=False)
    monkeypatch.setenv("HH_POLICY_W", "0")   # the tile forms only: left alone, 0 hands a batch this large to hh_k_policy_w16 (last bits differ)
    R = 16384 + 64   # 0 = by row count
    obs = torch.from_numpy(rng.random((R, 30)).astype(np.float32)).cuda()
    sel = torch.from_numpy(sels[rng.integers(1, len(sels), R)]).cuda()
    bank = _bank(7, max_rows=R)
    outs = []
    for rows in (0, 32, 64):
        bank.set_tile_rows(rows)
        lg = torch.zeros((R, 32), device="cuda")
        act = bank.act(obs, sel, logits=lg).clone()
        torch.cuda.synchronize()
        outs.append((lg, act))
    assert all(torch.equal(lg, outs[0][0]) and torch.equal(act, outs[0][1]) for lg, act in outs[1:])
    with pytest.raises(RuntimeError):
        bank.set_tile_rows(48)
    bank.close()


@pytest.mark.gpu
@pytest.mark.parametrize("force_w,nA,nO,N,form", [("0", 3, 3, 1003, "split-fp16"), ("2", 3, 3, 517, "split-fp16"), ("0", 2, 3, 300, "split-fp16"),
                                                  ("0", 3, 3, 700, "weights-through-lds-16-rows"), ("0", 3, 3, 900, "split-fp16-64-row-tiles")],
                         ids=["3v3", "3v3-W2", "2v3", "3v3-weights-through-lds", "3v3-64-row-tiles"])
def test_rows_binned_by_the_world_kernels_give_the_same_macro_steps(monkeypatch, force_w, nA, nO, N, form):
    """hh_bind_policy: the phase kernels write the bank's row lists themselves and hh_policy_act_binned runs the forward only
    (the last workgroup clears the counters; hh_hl_end drops what the last tick binned).  Same worlds, same weights: every macro
    step's outputs, the pilots' actions of every sub-step and the final state equal the self-contained form (binning pass from
    pilot_mode per call); afterwards the bank still serves selector calls, and unbinding restores the plain behaviour."""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_FORCE_W", force_w)
    _form(monkeypatch, form)
    kw = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=21, auto_reset=True, horizon=80)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    assert torch.equal(a.reset(), b.reset())
    pa, pb = pilots.NetPilot(a, seed=4, bind=False), pilots.NetPilot(b, seed=4, bind=True)
    log_a, log_b = [], []

    def tap(pilot, log):
        def f(po, pm):
            act = pilot(po, pm)
            log.append((act.clone(), pm.clone()))
            return act
        return f
    rng = np.random.default_rng(3)
    dones = 0
    for step in range(6):
        cmd = torch.from_numpy(rng.integers(0, 3, (N, nA)).astype(np.int8)).cuda()
        outs_a = macro_step(a, cmd, tap(pa, log_a))
        outs_b = macro_step(b, cmd, tap(pb, log_b))
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        dones += int(outs_a[3].sum())
    assert len(log_a) == len(log_b) == 6 * 32
    for k, ((xa, ma), (xb, mb)) in enumerate(zip(log_a, log_b)):
        assert torch.equal(ma, mb), f"call {k}: selector bytes"
        live = ma != 0
        assert torch.equal(xa[live], xb[live]), f"call {k}: pilots' actions"
    sa, sb = a.get_state(), b.get_state()
    for key in sa:
        assert np.array_equal(sa[key], sb[key]), key
    assert dones > 0
    # the bound bank still answers a plain selector call, and an unbound pilot on the same world behaves like the other world's
    obs = torch.rand((64, 30), device="cuda")
    sel = torch.full((64,), pilots.SEL_FIGHT1, dtype=torch.uint8, device="cuda")
    assert torch.equal(pb.bank.act(obs, sel), pa.bank.act(obs, sel))
    pb.close()
    cmd = torch.from_numpy(rng.integers(0, 3, (N, nA)).astype(np.int8)).cuda()
    for x, y in zip(macro_step(a, cmd, pa), macro_step(b, cmd, pb)):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_binding_survives_either_side_going_away():
    """hh_bind_policy keeps raw device pointers of the bank inside the world: destroying the bank, rebinding it to another world or
    destroying the world first must all leave both sides usable"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    kw = dict(n_arenas=64, env_kind=1, seed=2, auto_reset=True)
    w1, w2 = World(make_config(**kw)), World(make_config(**kw))
    w1.reset(), w2.reset()
    cmd = torch.ones((64, 3), dtype=torch.int8, device="cuda")
    bank = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=1, max_rows=64 * 6)
    p1 = pilots.NetPilot(w1, bank=bank)                  # bound to w1
    ref = [t.clone() for t in macro_step(w1, cmd, p1)]
    p2 = pilots.NetPilot(w2, bank=bank)                  # the binding moves to w2: w1 is back to emitting selector bytes only
    out2 = macro_step(w2, cmd, p2)
    for x, y in zip(ref, out2):
        assert torch.equal(x, y)
    p1.world = None                                       # w1's pilot falls back to the binning pass (its world is no longer bound)
    macro_step(w1, cmd, p1)
    bank.close()                                          # bank destroyed while bound to w2
    tape = torch.zeros((16, 64, 6, 4), dtype=torch.int8, device="cuda")
    w2.hl_rollout(cmd, tape)                              # w2 still steps (no dangling list pointers)
    po, pm = w2.hl_begin(cmd)
    torch.cuda.synchronize()
    assert pm.any()
    bank2 = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=1, max_rows=64 * 6)
    w2.bind_policy(bank2)
    w2.close()                                            # world destroyed while bound
    obs = torch.rand((64, 30), device="cuda")
    sel = torch.full((64,), pilots.SEL_FIGHT2, dtype=torch.uint8, device="cuda")
    assert bank2.act(obs, sel).shape == (64, 4)           # the bank lives on
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [4, 5])
def test_opponent_rows_binned_by_step_begin_give_the_same_steps(level):
    """hh_bind_policy on a LowLevelEnv world: hh_step_begin bins the frozen opponents' rows itself (selector from the aircraft type and
    the arena's level-5 draw).  Same worlds and weights: every step equals the form that builds the selector bytes in torch and runs
    the binning pass (a shared bank is never bound)"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.world import World, make_config
    N = 3000
    kw = dict(n_arenas=N, level=level, seed=17, auto_reset=True, ext_opp_actions=True, horizon=40)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    assert torch.equal(a.reset(), b.reset())
    shared = pilots.PolicyBank.random_init(torch.device("cuda", 0), seed=6, max_rows=N * 2)
    na, nb = pilots.OpponentNets(a, bank=shared), pilots.OpponentNets(b, seed=6)      # selector bytes + binning pass | bound
    rng = np.random.default_rng(2)
    mode = -1 if level == 5 else 0
    dones = 0
    for t in range(60):
        act = torch.from_numpy(random_actions(rng, (N,), 2)).cuda()
        oa, ob = a.step_begin(act, mode), b.step_begin(act, mode)
        assert torch.equal(oa, ob)
        xa, xb = na(oa).clone(), nb(ob).clone()
        live = oa.abs().sum(dim=2) > 0
        assert torch.equal(xa[live], xb[live]), f"t={t}: opponents' actions"
        outs_a, outs_b = a.step_finish(xa), b.step_finish(xb)
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"t={t}: {name}"
        dones += int(outs_a[3].sum())
    sa, sb = a.get_state(), b.get_state()
    for key in sa:
        assert np.array_equal(sa[key], sb[key]), key
    assert dones > N // 4
    if level == 5:
        assert set(np.unique(a.opp_policy().cpu().numpy())) == {3, 4, 5}


def _stub_reference_module(kind, seed):
    from helpers import stub_reference_module
    return stub_reference_module(kind, seed)


@pytest.mark.parametrize("kind", [PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2], ids=lambda k: PN.KIND_NAMES[k])
def test_weights_of_a_loaded_reference_module_are_recognised(kind):
    """PolicyBank.from_modules takes what the reference torch.load()s (env_base.py:312-347): architecture from the parameter shapes,
    actor tensors by their state_dict() names"""
    net, sd = _stub_reference_module(kind, 5)
    got_kind, got = PN.from_torch_module(net)
    assert got_kind == kind and set(got) == set(PN.actor_keys(kind))
    for k in sd:
        assert np.array_equal(got[k], sd[k])


@pytest.mark.gpu
def test_bank_from_loaded_modules_acts_like_the_torch_forward():
    from hhmarl_2d_amd import pilots
    mods = {s: _stub_reference_module(kind, 8)[0] for s, kind in enumerate((PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2))}
    bank = pilots.PolicyBank.from_modules(torch.device("cuda", 0), mods, max_rows=4096)
    bank.set_lut({pilots.SEL_FIGHT1: 0, pilots.SEL_FIGHT2: 1, pilots.SEL_ESC1: 2, pilots.SEL_ESC2: 3})
    rng = np.random.default_rng(1)
    obs = torch.from_numpy(rng.random((4096, 30)).astype(np.float32)).cuda()
    sels = np.array([pilots.SEL_FIGHT1, pilots.SEL_FIGHT2, pilots.SEL_ESC1, pilots.SEL_ESC2], dtype=np.uint8)
    sel = torch.from_numpy(sels[rng.integers(0, 4, 4096)]).cuda()
    logits = torch.zeros((4096, 32), device="cuda")
    bank.act(obs, sel, logits=logits)
    for byte, kind in zip(sels, (PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2)):
        idx = (sel == int(byte)).nonzero().flatten()
        ref = PR.torch_forward(kind, PN.random_weights(kind, 8), obs[idx].cpu())
        assert (logits[idx, : PN.N_OUT[kind]].cpu() - ref).abs().max() <= LOGIT_TOL


def _write_policy_dir(tmp_path, seed_of):
    """exported-policy files with the reference's names (env_base.py:312-347), synthetic weights"""
    for name, (kind, seed) in seed_of.items():
        torch.save(_stub_reference_module(kind, seed)[0], os.path.join(tmp_path, name))


@pytest.mark.gpu
def test_facades_load_the_reference_policy_files_themselves(tmp_path):
    """env_config["policy_dir"]: the facades do what the reference's _get_policies does — pick the exported policies by file name for
    the level / mode / env kind — and fly them; level 5: each arena's opponents fly the set its per-episode draw names"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.config import make_args
    from hhmarl_2d_amd.env_hetero import LowLevelEnv
    from hhmarl_2d_amd.env_hier import HighLevelEnv
    files = {"L3_AC1_fight.pt": (PN.FIGHT1, 31), "L3_AC2_fight.pt": (PN.FIGHT2, 32), "L4_AC1_fight.pt": (PN.FIGHT1, 41), "L4_AC2_fight.pt": (PN.FIGHT2, 42),
             "L3_AC1_escape.pt": (PN.ESC1, 33), "L3_AC2_escape.pt": (PN.ESC2, 34), "L5_AC1_fight.pt": (PN.FIGHT1, 51), "L5_AC2_fight.pt": (PN.FIGHT2, 52)}
    _write_policy_dir(str(tmp_path), files)
    n = 300
    env = LowLevelEnv({"args": make_args(0, level=5, horizon=25), "num_envs": n, "seed": 4, "policy_dir": str(tmp_path)})
    env.reset()
    rng = np.random.default_rng(0)
    seen = set()
    for t in range(8):
        env.step({1: random_actions(rng, (n,), 1)[:, 0], 2: random_actions(rng, (n,), 1)[:, 0, :3]})
        seen |= set(np.unique(env.world.opp_policy().cpu().numpy()).tolist())
    assert seen == {3, 4, 5}
    # which weights fly: every opponent row's action equals the torch forward of the file its arena's draw names
    w = env.world
    act = torch.from_numpy(random_actions(rng, (n,), 2)).cuda()
    opp_obs = w.step_begin(act, -1)
    got = env.opponent_policy(opp_obs, env).clone()
    k = w.opp_policy().cpu().numpy()
    by_k = {3: ("L3_AC1_fight.pt", "L3_AC2_fight.pt"), 4: ("L4_AC1_fight.pt", "L4_AC2_fight.pt"), 5: ("L3_AC1_escape.pt", "L3_AC2_escape.pt")}
    checked = 0
    for kk, names in by_k.items():
        for slot, name in enumerate(names):
            kind, seed = files[name]
            idx = np.nonzero(k == kk)[0]
            idx = idx[(opp_obs[idx, slot].abs().sum(dim=1) > 0).cpu().numpy()]
            if len(idx) == 0:
                continue
            ref = PR.torch_forward(kind, PN.random_weights(kind, seed), opp_obs[idx, slot].cpu())
            parts = ref.split(PN.ACTION_SPLIT[: 4 if PN.N_OUT[kind] == 26 else 3], dim=1)
            clear = torch.stack([(p.topk(2, dim=1).values[:, 0] - p.topk(2, dim=1).values[:, 1]) > 1e-4 for p in parts], dim=1).all(dim=1)
            assert torch.equal(got[idx, slot].cpu()[clear], PR.decode(ref, PN.N_OUT[kind])[clear]), (kk, name)
            checked += int(clear.sum())
    assert checked > n
    env.close()
    # HighLevelEnv: L{eval_level_ag} fights + L5 escapes, falling back to the L3 escapes when those were not exported
    hl = HighLevelEnv({"args": make_args(1, eval_level_ag=5), "num_envs": 64, "seed": 2, "policy_dir": str(tmp_path)})
    assert isinstance(hl.pilot, pilots.VariantNetPilot) and sorted(hl.pilot.bank.kinds.values()) == [PN.FIGHT1, PN.FIGHT2, PN.ESC1, PN.ESC2]   # the facade's default pilot-row form
    hl.reset()
    obs, rew, term, trunc, info = hl.step({1: np.ones(64, dtype=np.int64), 2: np.zeros(64, dtype=np.int64), 3: np.full(64, 2)})
    assert obs[1].shape == (64, 34) and np.isfinite(obs[1]).all()
    hl.close()
    hl = HighLevelEnv({"args": make_args(1, eval_level_ag=5), "num_envs": 4, "seed": 2, "policy_dir": str(tmp_path), "pilot_rows": "sides"})
    assert isinstance(hl.pilot, pilots.NetPilot)
    hl.close()
    with pytest.raises(FileNotFoundError):
        HighLevelEnv({"args": make_args(1, eval_level_ag=4), "num_envs": 4, "policy_dir": str(tmp_path / "missing")})
