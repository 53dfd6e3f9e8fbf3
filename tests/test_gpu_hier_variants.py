"""The variant-row form of the commander step with the pilot networks in the loop (include/hh_abi.h: hh_hl_begin_variants / hh_hl_act_tick —
ONE launch and ONE policy call per sub-step instead of two and two) flies the same trajectories as the standard phase path: the reference lets the
opponents' pilots observe the agents' weapon flags of the same sub-step (env_base.py:208-211), an agent's action can only raise its flag, so every
opponent's row is evaluated in its up-to-four variants ahead of the agents' action and the matching one is used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(kw, monkeypatch):
    from hhmarl_2d_amd.pilots import NetPilot, PolicyBank, VariantNetPilot
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_POLICY_W", "0")       # both banks on the tile form of the same width: a row's logits do not depend on its tile
    monkeypatch.setenv("HH_POLICY_TILE", "32")
    a, b = World(make_config(**kw)), World(make_config(**kw))
    pa = NetPilot(a, PolicyBank.random_init(a.device, seed=5, max_rows=a.N * 6))
    pb = VariantNetPilot(b, PolicyBank.random_init(b.device, seed=5, max_rows=b.N * 15))
    return a, b, pa, pb


@pytest.mark.parametrize("kw", [dict(n_arenas=257), dict(n_arenas=64, n_agents=2, n_opps=3, hier_opp_fight_ratio=0, horizon=150),
                                dict(n_arenas=100, n_agents=3, n_opps=1, glob_frac=0.3, hier_opp_fight_ratio=100, friendly_kill=False),
                                dict(n_arenas=40, hier_opp_fight_ratio=50, hier_action_assess=False, horizon=60)],
                         ids=["3v3", "2v3_escaping_opps", "3v1_fight", "3v3_mixed"])
def test_variant_rows_fly_the_standard_trajectories(kw, monkeypatch):
    import torch
    from hhmarl_2d_amd.env_hier import macro_step
    kw = dict(dict(env_kind=1, seed=31, arena_offset=7, auto_reset=True), **kw)
    a, b, pa, pb = _pair(kw, monkeypatch)
    assert torch.equal(a.reset(), b.reset())
    rng = np.random.default_rng(11)
    N, nA = kw["n_arenas"], a.n_agents
    used = np.zeros(4, dtype=np.int64)
    for step in range(30):
        cmd = torch.from_numpy(rng.integers(0, 3, (N, nA)).astype(np.int8)).cuda()
        if step % 5 == 0:   # phase by phase: every row the standard path hands its pilots is a row of the variant buffer, every action equal
            po, pm = a.hl_begin(cmd)
            vo, vm = b.hl_begin_variants(cmd)
            for sub in range(16):
                act = pa(po, pm).clone()
                vact = pb(vo, vm).clone()
                live = pm[:, :nA] != 0
                assert torch.equal(pm[:, :nA], vm[:, :nA]) and torch.equal(po[:, :nA][live], vo[:, :nA][live]), f"{step}/{sub}: agents' rows"
                assert torch.equal(act[:, :nA][live], vact[:, :nA][live]), f"{step}/{sub}: agents' actions"
                po, pm = a.hl_agents_act(act)
                act_o = pa(po, pm)
                # the opponents' rows of the standard path (after the agents acted) are among the variants
                ovo = vo[:, 3:].reshape(N, 3, 4, 30)
                ovm = vm[:, 3:].reshape(N, 3, 4)
                for j in range(a.n_units - nA):   # the standard buffer keeps opponent j in slot n_agents + j
                    on = pm[:, nA + j] != 0
                    if not bool(on.any()):
                        continue
                    match = (ovo[:, j] == po[:, nA + j][:, None, :]).all(-1) & (ovm[:, j] == pm[:, nA + j][:, None])
                    assert bool(match[on].any(-1).all()), f"{step}/{sub}: opponent {j}: its row is not among the variants"
                    first = match[on].float().argmax(-1).cpu().numpy()
                    used += np.bincount(first, minlength=4)
                act[:, nA:] = act_o[:, nA:]
                po, pm, ra = a.hl_tick(act)
                vo, vm, rb = b.hl_act_tick(vact)
                assert ra == rb, f"{step}/{sub}: running"
                assert np.array_equal(a.event_masks(), b.event_masks()), f"{step}/{sub}: event masks"
                if ra == 0:
                    break
            outs_a, outs_b = a.hl_end(), b.hl_end()
        else:
            outs_a = macro_step(a, cmd, pa, early_exit=True)
            outs_b = macro_step(b, cmd, pb, early_exit=True)
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        sa, sb = a.get_state(), b.get_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), f"step {step}: state {k}"
        for x, y in zip(a.eval_info(), b.eval_info()):
            assert torch.equal(x, y)
        assert a.hl_tick_count() == b.hl_tick_count()
    for x, y in zip(a.episode_stats(), b.episode_stats()):
        assert torch.equal(x, y)
    assert not a.action_faults().any() and not b.action_faults().any()
    if kw.get("hier_opp_fight_ratio", 75) not in (0,):
        assert used[1:].sum() > 0, "no opponent ever needed a variant: the test never exercised the point"
    pa.close(); pb.close()


def test_variant_rows_unbound_and_refusals(monkeypatch):
    """without hh_bind_policy the selector bytes [N, 15] drive a binning pass (same actions); a bank too small for 15 rows per arena is refused;
    ten-slot worlds are refused"""
    import torch
    from hhmarl_2d_amd.pilots import PolicyBank, VariantNetPilot
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_POLICY_W", "0")
    monkeypatch.setenv("HH_POLICY_TILE", "32")
    kw = dict(n_arenas=50, env_kind=1, seed=3, arena_offset=0, auto_reset=True)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    pa = VariantNetPilot(a, PolicyBank.random_init(a.device, seed=9, max_rows=50 * 15), bind=True)
    pb = VariantNetPilot(b, PolicyBank.random_init(b.device, seed=9, max_rows=50 * 15), bind=False)
    a.reset(); b.reset()
    from hhmarl_2d_amd.env_hier import macro_step
    rng = np.random.default_rng(0)
    for step in range(6):
        cmd = torch.from_numpy(rng.integers(0, 3, (50, 3)).astype(np.int8)).cuda()
        for x, y in zip(macro_step(a, cmd, pa), macro_step(b, cmd, pb)):
            assert torch.equal(x, y)
    small = PolicyBank.random_init(a.device, seed=9, max_rows=50 * 6)
    c = World(make_config(**kw)); c.reset(); c.bind_policy(small)
    with pytest.raises(Exception):
        c.hl_begin_variants(torch.zeros((50, 3), dtype=torch.int8, device=c.device))
    c.bind_policy(None)
    wide = World(make_config(n_arenas=4, env_kind=1, n_agents=4, n_opps=4)); wide.reset()
    with pytest.raises(Exception):
        wide.hl_begin_variants(torch.zeros((4, 4), dtype=torch.int8, device=wide.device), (torch.zeros((4, 15, 30), device=wide.device), torch.zeros((4, 15), dtype=torch.uint8, device=wide.device)))


@pytest.mark.parametrize("form", ["3", "2"], ids=["w16x8", "w16x4"])
def test_streamed_policy_forms_walk_their_tiles_when_the_row_estimate_is_low(form, monkeypatch):
    """hh_policy_act_binned_live sizes the grid of hh_k_policy_w16 by the caller's ESTIMATE of the listed rows (hhp_launch_forward): the workgroups walk the
    tiles grid-stride, so a call with many more rows than estimated returns the same actions and logits as a call whose grid covers every row slot."""
    import torch
    from hhmarl_2d_amd.pilots import PolicyBank, VariantNetPilot
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_POLICY_W", form)
    rng = np.random.default_rng(2)
    got = []
    for est in (None, 1, 700):   # a grid for every slot; a grid of a handful of workgroups for the same ~6 000 listed rows; one that is short by a few tiles
        w = World(make_config(n_arenas=600, env_kind=1, seed=3, auto_reset=True))
        w.reset()
        p = VariantNetPilot(w, PolicyBank.random_init(w.device, seed=5, max_rows=w.N * 15))
        cmd = torch.from_numpy(np.random.default_rng(2).integers(0, 3, (w.N, 3)).astype(np.int8)).cuda()
        rows = w.N * 15
        po, pm = w.hl_begin_variants(cmd)
        assert int((pm != 0).sum()) > 5000
        act = torch.full((w.N, 15, 4), 77, dtype=torch.int8, device=po.device)
        logits = torch.zeros((rows, 32), dtype=torch.float32, device=po.device)
        p.bank.act_binned_live(po, act, rows if est is None else est, logits=logits)
        torch.cuda.synchronize()
        m = (pm != 0)
        got.append((pm.clone(), act[m].clone(), logits.view(w.N, 15, 32)[m].clone()))
        p.close()
        w.close()
    for pm, a, l in got[1:]:
        assert torch.equal(pm, got[0][0]) and torch.equal(a, got[0][1]) and torch.equal(l, got[0][2])


def test_sub_worlds_replayed_from_their_own_graphs_equal_one_world(monkeypatch):
    """The form bench.py measures the commander step with the networks in: the arenas split into four sub-worlds (disjoint global arena ids), each captured in
    its own HIP graph and replayed on its own stream with no join between commander steps.  Every commander step's outputs and the final state equal those of
    ONE world of all the arenas stepped eagerly on the default stream."""
    import torch
    import bench
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.pilots import PolicyBank, VariantNetPilot
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_POLICY_W", "3")   # one forward form on both sides (a row's logits do not depend on the tile it rides in, only on the form)
    N, K, STEPS = 512, 4, 6
    n = N // K
    kw = dict(env_kind=1, seed=17, auto_reset=True)
    one = World(make_config(n_arenas=N, arena_offset=4000, **kw))
    subs = [World(make_config(n_arenas=n, arena_offset=4000 + k * n, **kw)) for k in range(K)]
    obs0 = one.reset()
    for k, w in enumerate(subs):
        assert torch.equal(w.reset(), obs0[k * n:(k + 1) * n])
    p_one = VariantNetPilot(one, PolicyBank.random_init(one.device, seed=5, max_rows=N * 15))
    p_sub = [VariantNetPilot(w, PolicyBank.random_init(w.device, seed=5, max_rows=n * 15)) for w in subs]
    rng = np.random.default_rng(3)
    cmds = torch.from_numpy(rng.integers(0, 3, (STEPS, N, 3)).astype(np.int8)).cuda()
    # one world, eager
    want = []
    for s in range(STEPS):
        want.append([x.clone() for x in macro_step(one, cmds[s].contiguous(), p_one)])
    torch.cuda.synchronize()
    # four sub-worlds, one graph each, pipelined: a step's outputs are copied aside on the sub-world's own stream
    streams = bench.make_streams(torch, K)
    cmd_static = [cmds[0, k * n:(k + 1) * n].clone() for k in range(K)]
    outs = [w.alloc_outputs() for w in subs]
    pbufs = [w.alloc_pilot_variants() for w in subs]
    graphs = []
    for k in range(K):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=(streams[k] if streams[k].cuda_stream != 0 else torch.cuda.Stream())):
            macro_step(subs[k], cmd_static[k], p_sub[k], out=outs[k], pilot_buf=pbufs[k])
        graphs.append(g)
    got = [[None] * K for _ in range(STEPS)]
    cur = torch.cuda.current_stream()
    for k in range(K):
        streams[k].wait_stream(cur)
    for s in range(STEPS):
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                cmd_static[k].copy_(cmds[s, k * n:(k + 1) * n])
                graphs[k].replay()
                got[s][k] = [x.clone() for x in outs[k]]
    for k in range(K):
        cur.wait_stream(streams[k])
    torch.cuda.synchronize()
    for s in range(STEPS):
        for i, name in enumerate(("obs", "reward", "valid", "done")):
            g_ = torch.cat([got[s][k][i] for k in range(K)], dim=0)
            assert torch.equal(g_, want[s][i]), f"commander step {s}: {name} of the four sub-worlds differs from the one world's"
    so = one.get_state()
    for k, w in enumerate(subs):
        sk = w.get_state()
        for key in so:
            assert np.array_equal(sk[key], so[key][k * n:(k + 1) * n]), f"final state {key} of sub-world {k}"
    for p in p_sub + [p_one]:
        p.close()
    for w in subs + [one]:
        w.close()
