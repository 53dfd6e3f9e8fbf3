"""HighLevelEnv (3-vs-3 commander, envs/env_hier.py) on the GPU: bit-exact against the CPU oracle with
random commander/pilot actions, and against the golden traces recorded from the real reference."""
import numpy as np
import pytest

from helpers import cfg_kwargs_from_meta, golden_files, load_golden, random_actions

pytestmark = pytest.mark.gpu


def _same_state(a, b, what):
    for k in ("ac_i", "rk_i", "ar_i", "tgt_id"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs at {np.argwhere(a[k] != b[k])[:5].tolist()}"
    for k in ("ac_f", "rk_f", "tgt_d"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} max diff {np.abs(a[k] - b[k]).max()}"


@pytest.mark.parametrize("kw", [dict(), dict(glob_frac=0.3, hier_opp_fight_ratio=50, hier_action_assess=False),
                                dict(friendly_kill=False, horizon=120)], ids=["default", "share", "nofriendly"])
def test_macro_step_parity(oracle, kw):
    import torch
    from hhmarl_2d_amd.world import World, make_config
    N = 170  # 10 arenas per 64-lane workgroup: 17 full groups; the partial-group case is covered by the traces (N = 1)
    base = dict(n_arenas=N, env_kind=1, seed=21, arena_offset=500, auto_reset=True)
    base.update(kw)
    g = World(make_config(**base))
    o = oracle.OracleWorld(oracle.make_config(**base))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    _same_state(g.get_state(), o.get_state(), "reset")
    rng = np.random.default_rng(4)
    dones = kills = 0
    for step in range(45):
        cmd = rng.integers(0, 3, (N, 3)).astype(np.int8)
        po, pm = g.hl_begin(torch.from_numpy(cmd).cuda())
        o.hl_begin(cmd)
        for sub in range(16):
            po_o, pm_o = o.hl_pilot_obs(0)
            assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), f"{step}/{sub}: agent pilot obs"
            act = random_actions(rng, (N,), 6)
            if step % 3 == 0:  # engage: everybody keeps the trigger pulled
                act[..., 2] = 1
            ta = torch.from_numpy(act).cuda()
            po, pm = g.hl_agents_act(ta)
            o.hl_agents_act(act)
            po_o, pm_o = o.hl_pilot_obs(1)
            assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), f"{step}/{sub}: opp pilot obs"
            po, pm, running = g.hl_tick(ta)
            running_o = o.hl_tick(act)
            assert running == running_o, f"{step}/{sub}: running {running} vs {running_o}"
            assert np.array_equal(g.event_masks(), o.event_masks()), f"{step}/{sub}: event masks"
            kills += int(np.count_nonzero(o.event_masks() & 0xFFFF))
            if running == 0:
                break
        outs = [x.cpu().numpy() for x in g.hl_end()]
        outs_o = o.hl_end()
        for a, b, name in zip(outs, outs_o, ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b), f"step {step}: {name}"
        dones += int(outs[3].sum())
        _same_state(g.get_state(), o.get_state(), f"step {step}")
        for a, b, name in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info(), ("last", "total")):   # env_base.py:91-107 on the device
            assert np.array_equal(a, b), f"step {step}: eval_info {name}"
        st = g.arena_status().cpu().numpy()
        assert np.array_equal(st[:, :3], o.get_state()["ar_i"][:, :3]), f"step {step}: arena status"
    for a, b in zip([x.cpu().numpy() for x in g.episode_stats()], o.episode_stats()):
        assert np.array_equal(a, b)
    assert dones > 0 and kills > 0
    tot = g.eval_info(clear_total=True)[1]
    assert int(tot[:, :3].sum()) > 0 and int(tot[:, 7].sum()) > 0 and int(g.eval_info()[1].sum()) == 0   # sums were cleared


@pytest.mark.parametrize("N,force_w", [(8192, "0"), (12003, "0"), (173, "2")],
                         ids=["configs3-8192-W1", "12003-auto-W2-partial-group", "173-forced-W2"])
def test_macro_step_parity_at_size(oracle, monkeypatch, N, force_w):
    """BASELINE configs[3] size (8192 arenas: hh_k_hier<6,64,1>) and the two-waves-per-SIMD instance hh_k_hier<6,64,2> that
    the host picks above one workgroup per SIMD (N = 12003: 1201 workgroups, the last one partially filled; and forced at a
    small size), >= 3 commander steps with every sub-step's pilot observations, outputs and the final state against the oracle"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_FORCE_W", force_w)
    base = dict(n_arenas=N, env_kind=1, seed=77, arena_offset=3, auto_reset=True, horizon=40)   # short horizon: resets inside the run
    g = World(make_config(**base))
    assert g.kernel_name() == ("hh_k_hier<6,64,2>" if (force_w == "2" or N > 10240) else "hh_k_hier<6,64,1>")
    o = oracle.OracleWorld(oracle.make_config(**base))
    assert np.array_equal(g.reset().cpu().numpy(), o.reset())
    rng = np.random.default_rng(N)
    dones = 0
    for step in range(4):
        cmd = rng.integers(0, 3, (N, 3)).astype(np.int8)
        po, pm = g.hl_begin(torch.from_numpy(cmd).cuda())
        o.hl_begin(cmd)
        for sub in range(16):
            po_o, pm_o = o.hl_pilot_obs(0)
            assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), f"{step}/{sub}: agent pilot obs"
            act = random_actions(rng, (N,), 6)
            act[..., 2] = 1
            ta = torch.from_numpy(act).cuda()
            po, pm = g.hl_agents_act(ta)
            o.hl_agents_act(act)
            po_o, pm_o = o.hl_pilot_obs(1)
            assert np.array_equal(pm.cpu().numpy(), pm_o) and np.array_equal(po.cpu().numpy(), po_o), f"{step}/{sub}: opp pilot obs"
            po, pm, running = g.hl_tick(ta)
            assert running == o.hl_tick(act), f"{step}/{sub}: running"
            if running == 0:
                break
        outs = [x.cpu().numpy() for x in g.hl_end()]
        for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b), f"step {step}: {name}"
        dones += int(outs[3].sum())
    _same_state(g.get_state(), o.get_state(), "final")
    for a, b in zip([x.cpu().numpy() for x in g.episode_stats()], o.episode_stats()):
        assert np.array_equal(a, b)
    assert dones > 0
    ticks = g.hl_tick_count()
    assert 0 < ticks <= 4 * 16 * N


@pytest.mark.parametrize("nA,nO", [(2, 3), (3, 1), (1, 1), (3, 2)], ids=lambda v: str(v))
def test_n_vs_m_parity(oracle, nA, nO):
    """evaluation.py's n-vs-m scenarios (README.md:43 of the reference): 1..3 agents against 1..3 opponents in the six unit slots of
    the 3-vs-3 kernel, phase path and persistent macro step, against the oracle (which holds exactly n + m units)"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    N, nU = 300, nA + nO
    base = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=5, arena_offset=9, auto_reset=True, horizon=80)
    g, h = World(make_config(**base)), World(make_config(**base))
    o = oracle.OracleWorld(oracle.make_config(**base))
    assert g.A == 6 and g.n_units == nU and o.A == nU
    og = g.reset()
    assert np.array_equal(og.cpu().numpy(), o.reset()) and torch.equal(h.reset(), og)
    rng = np.random.default_rng(nA * 10 + nO)
    dones = 0
    for step in range(12):
        cmd = rng.integers(0, 3, (N, nA)).astype(np.int8)
        tape = random_actions(rng, (16, N), 6)
        tape[..., 2] = 1
        po, pm = g.hl_begin(torch.from_numpy(cmd).cuda())
        o.hl_begin(cmd)
        for sub in range(16):
            po_o, pm_o = o.hl_pilot_obs(0)
            assert np.array_equal(pm.cpu().numpy()[:, :nU], pm_o) and np.array_equal(po.cpu().numpy()[:, :nU], po_o), f"{step}/{sub}: agents"
            assert not pm[:, nU:].any()
            ta = torch.from_numpy(tape[sub]).cuda()
            po, pm = g.hl_agents_act(ta)
            o.hl_agents_act(np.ascontiguousarray(tape[sub][:, :nU]))
            po_o, pm_o = o.hl_pilot_obs(1)
            assert np.array_equal(pm.cpu().numpy()[:, :nU], pm_o) and np.array_equal(po.cpu().numpy()[:, :nU], po_o), f"{step}/{sub}: opponents"
            po, pm, running = g.hl_tick(ta)
            assert running == o.hl_tick(np.ascontiguousarray(tape[sub][:, :nU])), f"{step}/{sub}"
        outs = [x.cpu().numpy() for x in g.hl_end()]
        for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b), f"step {step}: {name}"
        outs_h = h.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())       # the one-launch path
        for a, b, name in zip(outs, outs_h, ("obs", "reward", "valid", "done")):
            assert np.array_equal(a, b.cpu().numpy()), f"step {step}: {name} (hh_hl_rollout)"
        sg, so = g.get_state(), o.get_state()
        for k in sg:
            if k == "ar_i":
                assert np.array_equal(sg[k], so[k]), k
            else:
                assert np.array_equal(sg[k][:, :nU], so[k]), f"step {step}: {k}"
        for a, b in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info()):
            assert np.array_equal(a, b), f"step {step}: eval counters"
        dones += int(outs[3].sum())
    assert dones > 0


@pytest.mark.parametrize("N,force_w,horizon,apw", [(170, "0", 60, "0"), (8192, "0", 60, "0"), (12003, "0", 60, "0"), (333, "2", 60, "0"), (170, "0", 500, "0"),
                                                   (8192, "0", 500, "0"), (333, "2", 500, "0"), (170, "0", 60, "16"), (190, "0", 500, "16")],
                         ids=["170", "8192", "12003-W2", "333-forced-W2", "170-default-config", "configs3-8192-default-config", "333-forced-W2-default-config",
                              "170-ten-arenas-per-wave", "190-default-config-ten-arenas-per-wave"])
def test_persistent_macro_step_equals_phase_path(oracle, monkeypatch, N, force_w, horizon, apw):
    """hh_hl_rollout (one launch per commander step, actions from a resident tape) against the phase-by-phase path with the same
    tape: outputs, final state, event masks, eval counters, episode statistics and tick counts bit for bit; at N = 170 also
    against the oracle.  horizon = 500 is the reference's default HighLevelEnv configuration, which runs the macro-step instance
    compiled with that configuration as constants (hh_cfg_set_hl_default); any other horizon runs the general instance.  Worlds of up
    to 8192 arenas run 8 arenas per wave unless HH_APW=16 keeps the 10 that fill a wave."""
    import torch
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_FORCE_W", force_w)
    monkeypatch.setenv("HH_APW", apw)
    base = dict(n_arenas=N, env_kind=1, seed=8, arena_offset=11, auto_reset=True, horizon=horizon)
    a, b = World(make_config(**base)), World(make_config(**base))
    o = oracle.OracleWorld(oracle.make_config(**base)) if N <= 200 else None
    assert torch.equal(a.reset(), b.reset())
    if o is not None:
        o.reset()
    rng = np.random.default_rng(N)
    dones = 0
    for step in range(8):
        cmd_h = rng.integers(0, 3, (N, 3)).astype(np.int8)
        tape_h = random_actions(rng, (16, N), 6)
        tape_h[..., 2] |= (step % 2)            # every other step everybody keeps the trigger pulled
        cmd, tape = torch.from_numpy(cmd_h).cuda(), torch.from_numpy(tape_h).cuda()
        calls = [0]

        def pilot(po, pm):
            act = tape[(calls[0] // 2) % 16]
            calls[0] += 1
            return act
        outs_a = macro_step(a, cmd, pilot)
        outs_b = b.hl_rollout(cmd, tape)
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        assert np.array_equal(a.event_masks(), b.event_masks()), f"step {step}: event masks"
        _same_state(a.get_state(), b.get_state(), f"step {step}")
        for x, y in zip(a.eval_info(), b.eval_info()):
            assert torch.equal(x, y), f"step {step}: eval counters"
        assert a.hl_tick_count() == b.hl_tick_count(), f"step {step}: arena-ticks"
        dones += int(outs_b[3].sum())
        if o is not None:
            o.hl_begin(cmd_h)
            for k in range(16):
                o.hl_agents_act(tape_h[k])
                o.hl_tick(tape_h[k])
            for x, y, name in zip([t.cpu().numpy() for t in outs_b], o.hl_end(), ("obs", "reward", "valid", "done")):
                assert np.array_equal(x, y), f"step {step}: {name} vs oracle"
            _same_state(b.get_state(), o.get_state(), f"step {step} vs oracle")
    for x, y in zip(a.episode_stats(), b.episode_stats()):
        assert torch.equal(x, y)
    assert (dones > 0 or horizon > 60) and b.hl_tick_count() < 8 * 16 * N   # arenas do leave their macro step early


@pytest.mark.parametrize("path", golden_files("high"), ids=lambda p: p.split("env_")[-1][:-4])
def test_reference_traces_on_gpu(path):
    import torch
    from hhmarl_2d_amd.world import World, make_config
    g, meta = load_golden(path)
    w = World(make_config(**cfg_kwargs_from_meta(meta)))
    nA, ptr, nU = w.n_agents, 0, w.n_units      # nU < A for the n-vs-m traces: the remaining unit slots are never alive
    for r in range(len(g["kind"])):
        if g["kind"][r] == 0:
            obs = w.reset().cpu().numpy()[0]
            rew, val, done = np.zeros(nA), np.zeros(nA, dtype=np.uint8), 0
        else:
            po, pm = w.hl_begin(torch.from_numpy(np.ascontiguousarray(g["cmd"][r][None])).cuda())
            for k in range(g["nsub"][r]):
                a6 = np.zeros((1, w.A, 4), dtype=np.int8)   # six unit slots, ten with more than three aircraft on a side
                a6[0, :nU] = g["sub_act"][ptr]
                act = torch.from_numpy(a6).cuda()
                po1, pm1 = w.hl_agents_act(act)
                assert int((pm + pm1)[0, nU:].sum()) == 0 and float((po + po1)[0, nU:].abs().sum()) == 0.0
                mode = (pm + pm1).cpu().numpy()[0, :nU]
                pobs = (po + po1).cpu().numpy()[0, :nU]
                assert np.array_equal(mode & 3, g["sub_mode"][ptr]), f"row {r} sub {k}"
                assert np.abs(pobs - g["sub_obs"][ptr]).max() <= 1e-6, f"row {r} sub {k}"
                po, pm, running = w.hl_tick(act)
                ptr += 1
                assert running == (1 if k < g["nsub"][r] - 1 else 0), f"row {r}: macro step length"
            o, rw, v, d = [x.cpu().numpy() for x in w.hl_end()]
            obs, rew, val, done = o[0], rw[0], v[0], d[0]
        st = w.get_state()
        assert np.array_equal(st["tgt_id"][0][:nU], g["tgt_id"][r]) and np.array_equal(st["rk_i"][0][:nU], g["rk_i"][r]), f"row {r}"
        assert np.array_equal(st["ac_i"][0][:nU, :9], g["ac_i"][r][:, :9]) and not st["ac_i"][0][nU:, 0].any(), f"row {r}"
        assert np.array_equal(val, g["valid"][r]) and done == g["done"][r], f"row {r}"
        assert np.abs(st["ac_f"][0][:nU] - g["ac_f"][r]).max() <= 1e-9 and np.abs(obs - g["obs"][r]).max() <= 1e-6, f"row {r}"
        assert np.abs(rew - g["reward"][r]).max() <= 1e-6, f"row {r}"


@pytest.mark.gpu
def test_random_highlevel_configurations_parity(oracle):
    """configuration fuzz for HighLevelEnv: 14 seeded random combinations of side sizes (n-vs-m), reward sharing, action assessment,
    opponents' fight ratio, horizon and arena count; the one-launch macro step (whatever instance the configuration selects) against
    the oracle stepping the same tape phase by phase: outputs, eval counters, event masks and the final state bit for bit"""
    import torch
    from hhmarl_2d_amd.world import World, make_config
    rng = np.random.default_rng(424242)
    for trial in range(14):
        nA, nO = (3, 3) if trial % 2 == 0 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        kw = dict(n_arenas=int(rng.choice([1, 9, 10, 11, 170, 1001])), env_kind=1, n_agents=nA, n_opps=nO, horizon=int(rng.choice([40, 120, 500])),
                  glob_frac=float(rng.choice([0.0, 0.0, 0.3])), hier_action_assess=bool(rng.integers(0, 2)),
                  hier_opp_fight_ratio=int(rng.choice([0, 50, 75, 100])), friendly_kill=bool(rng.integers(0, 4)),
                  seed=int(rng.integers(0, 1 << 30)), arena_offset=int(rng.integers(0, 1 << 20)), auto_reset=True)
        if trial % 4 == 0:   # the reference's default HighLevelEnv configuration: its own compiled instance
            kw.update(horizon=500, glob_frac=0.0, hier_action_assess=True, hier_opp_fight_ratio=75, friendly_kill=True)
        g, o = World(make_config(**kw)), oracle.OracleWorld(oracle.make_config(**kw))
        assert np.array_equal(g.reset().cpu().numpy(), o.reset()), (trial, kw)
        N, nU = kw["n_arenas"], nA + nO
        for step in range(4):
            cmd = rng.integers(0, 3, (N, nA)).astype(np.int8)
            tape = random_actions(rng, (16, N), 6)
            tape[..., 2] |= step % 2
            outs = [x.cpu().numpy() for x in g.hl_rollout(torch.from_numpy(cmd).cuda(), torch.from_numpy(tape).cuda())]
            o.hl_begin(cmd)
            for k in range(16):
                o.hl_agents_act(np.ascontiguousarray(tape[k][:, :nU]))
                o.hl_tick(np.ascontiguousarray(tape[k][:, :nU]))
            for a, b, name in zip(outs, o.hl_end(), ("obs", "reward", "valid", "done")):
                assert np.array_equal(a, b), (trial, step, name, kw)
            for a, b in zip([x.cpu().numpy() for x in g.eval_info()], o.eval_info()):
                assert np.array_equal(a, b), (trial, step, "eval counters", kw)
            assert np.array_equal(g.event_masks(), o.event_masks()), (trial, step, kw)
        sg, so = g.get_state(), o.get_state()
        for k in sg:   # the world keeps six unit slots, the oracle exactly n + m units
            assert np.array_equal(sg[k] if k == "ar_i" else sg[k][:, :nU], so[k]), (trial, k, kw)
