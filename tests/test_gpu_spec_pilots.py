"""hh_hl_set_speculation (pilots.NetPilot(speculate=True)): the commander step with the pilot networks in the loop where ONE policy call serves both
sides of a sub-step.  An opponent's pilot observes, after the agents acted (env_hier.py:126-133 in unit id order), what it would have observed before
except for the agents' weapon flags (env_base.py:208-211) — so hh_hl_begin / hh_hl_tick emit the opponents' rows ahead of time and hh_hl_agents_act
re-lists only the opponents of arenas in which an agent's flag did change.  Same actions, outputs and state as the one-side-at-a-time order, bit for
bit; and the re-listing is what makes it so (HH_SPEC_NO_REDO: the same run without it diverges)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_state(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: {k}"


@pytest.mark.parametrize("N,nA,nO,force_w", [(2048, 3, 3, "0"), (203, 3, 3, "0"), (640, 2, 3, "0"), (1500, 3, 1, "2")],
                         ids=["2048-3v3", "203-3v3", "640-2v3", "1500-3v1-W2"])
def test_speculative_pilot_rows_give_the_same_commander_steps(monkeypatch, N, nA, nO, force_w):
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    monkeypatch.setenv("HH_FORCE_W", force_w)
    # ONE policy kernel form for every call of both worlds (the forms agree to the last bits of the logits only: an arg-max on a near-tie may differ between
    # them, and the two orders issue calls of different sizes): what is compared here is the order of evaluation, not the forms
    monkeypatch.setenv("HH_POLICY_W", "0")
    monkeypatch.setenv("HH_POLICY_TILE", "32")
    kw = dict(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=33, auto_reset=True, horizon=60, arena_offset=9)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    assert torch.equal(a.reset(), b.reset())
    pa, pb = pilots.NetPilot(a, seed=6), pilots.NetPilot(b, seed=6, speculate=True)
    log_a, log_b = [], []

    def tap(pilot, log, keep):
        def f(po, pm, **kw_):
            act = pilot(po, pm, **kw_)
            if keep(len(log)):
                log.append((act.clone(), pm.clone()))
            else:
                log.append(None)
            return act
        f.speculative = getattr(pilot, "speculative", False)
        return f
    rng = np.random.default_rng(4)
    dones = fired = 0
    for step in range(6):
        cmd = torch.from_numpy(rng.integers(0, 3, (N, nA)).astype(np.int8)).cuda()
        k0 = len(log_a)
        outs_a = macro_step(a, cmd, tap(pa, log_a, lambda k: True))
        outs_b = macro_step(b, cmd, tap(pb, log_b, lambda k: True))
        for x, y, name in zip(outs_a, outs_b, ("obs", "reward", "valid", "done")):
            assert torch.equal(x, y), f"step {step}: {name}"
        assert np.array_equal(a.event_masks(), b.event_masks()), f"step {step}: event masks"
        for x, y in zip(a.eval_info(), b.eval_info()):
            assert torch.equal(x, y), f"step {step}: eval counters"
        # the actions the world consumed: after the second pilot call of a sub-step both buffers hold every live unit's final action
        for k in range(k0 + 1, len(log_a), 2):
            (xa, ma), (xb, _) = log_a[k], log_b[k]
            live = ma != 0                      # the opponents' selector bytes (one-side order)
            assert torch.equal(xa[live], xb[live]), f"call {k}: opponents' actions"
        for k in range(k0, len(log_a), 2):
            (xa, ma), (xb, mb) = log_a[k], log_b[k]
            live = ma != 0                      # the agents' rows
            assert torch.equal(xa[live], xb[live]) and torch.equal(mb[live], ma[live]), f"call {k}: agents' actions"
            fired += int(xa[live][:, 2].sum())
        dones += int(outs_a[3].sum())
    _same_state(a.get_state(), b.get_state(), "final")
    assert a.hl_tick_count() == b.hl_tick_count()
    assert dones > 0 and fired > 0
    pa.close(); pb.close()


def test_without_the_relisting_the_speculation_diverges(monkeypatch):
    """the test above is sensitive to what it tests: with the re-listing switched off (HH_SPEC_NO_REDO=1, a test-only knob) the opponents act on rows that
    miss the agents' same-sub-step weapon flags and the worlds drift apart"""
    from hhmarl_2d_amd import pilots
    from hhmarl_2d_amd.env_hier import macro_step
    from hhmarl_2d_amd.world import World, make_config
    N = 2048
    kw = dict(n_arenas=N, env_kind=1, seed=33, auto_reset=True, horizon=60, arena_offset=9)
    a, b = World(make_config(**kw)), World(make_config(**kw))
    a.reset(); b.reset()
    monkeypatch.setenv("HH_POLICY_W", "0")
    monkeypatch.setenv("HH_POLICY_TILE", "32")
    pa = pilots.NetPilot(a, seed=6)
    monkeypatch.setenv("HH_SPEC_NO_REDO", "1")
    pb = pilots.NetPilot(b, seed=6, speculate=True)
    rng = np.random.default_rng(4)
    differs = False
    for step in range(6):
        cmd = torch.from_numpy(rng.integers(0, 3, (N, 3)).astype(np.int8)).cuda()
        outs_a, outs_b = macro_step(a, cmd, pa), macro_step(b, cmd, pb)
        differs = differs or any(not torch.equal(x, y) for x, y in zip(outs_a, outs_b))
    sa, sb = a.get_state(), b.get_state()
    differs = differs or any(not np.array_equal(sa[k], sb[k]) for k in sa)
    assert differs
    pa.close(); pb.close()
