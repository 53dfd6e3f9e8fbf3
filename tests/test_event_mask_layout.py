"""The event-mask layout (include/hh_spec.h: HH_EV_BIT) pinned on the CPU oracle: the units a mask names as killed by cannon / killed by rocket / out of bounds
are exactly the units that stopped existing in that sub-step, in the six-slot layout (8 bits per class) and in the ten-slot layout of arenas with more than three
aircraft on a side (10 bits per class, launches as one bit per side).  The GPU world is compared with these masks bit for bit in tests/test_gpu_hier*.py."""
import numpy as np
import pytest


@pytest.mark.parametrize("sides", [(3, 3), (2, 3), (5, 5), (4, 2), (1, 5)], ids=lambda s: f"{s[0]}v{s[1]}")
def test_masks_name_exactly_the_units_that_died(oracle, sides):
    nA, nO = sides
    nU, N = nA + nO, 60
    wide = max(nA, nO) > 3
    width, slots = (10, 10) if wide else (8, 6)
    o = oracle.OracleWorld(oracle.make_config(n_arenas=N, env_kind=1, n_agents=nA, n_opps=nO, seed=5, arena_offset=40, auto_reset=False, horizon=400))
    o.reset()
    rng = np.random.default_rng(1)
    deaths = launches = 0
    for step in range(12):
        o.hl_begin(rng.integers(0, 3, (N, nA)).astype(np.int8))
        for sub in range(16):
            before = o.get_state()["ac_i"][:, :, 0].copy()
            rk_before = o.get_state()["rk_i"][:, :, 0].copy()
            sub_before = o.hl_cmd()[1].copy()
            act = np.stack([rng.integers(0, 13, (N, nU)), rng.integers(0, 9, (N, nU)), np.ones((N, nU), dtype=np.int64), np.ones((N, nU), dtype=np.int64)], axis=-1).astype(np.int8)
            o.hl_agents_act(act)
            running = o.hl_tick(act)
            st = o.get_state()
            after, rk_after = st["ac_i"][:, :, 0], st["rk_i"][:, :, 0]
            em = o.event_masks().astype(np.uint64)
            ran = o.hl_cmd()[1] > sub_before   # arenas that ran this sub-step (the others keep the mask of their last one)
            died = (before == 1) & (after == 0)
            assert not died[~ran].any()
            em = np.where(ran, em, np.uint64(0))
            field = lambda c: (em[:, None] >> np.uint64(width * c)) >> np.arange(nU, dtype=np.uint64)[None, :] & np.uint64(1)
            named = (field(0) | field(1) | field(2)).astype(bool)
            assert np.array_equal(named, died), f"step {step}/{sub}: the masks' kill / out-of-bounds bits are not the units that died"
            assert not ((field(0) & field(1)).any()), "a unit killed by cannon AND by rocket in one sub-step"
            if wide:
                assert not (em >> np.uint64(32)).any()
                new_rk = (rk_before == 0) & (rk_after == 1)   # a rocket that exists now and did not before: launched in this sub-step (and still flying)
                side_bits = np.stack([(em >> np.uint64(30)) & np.uint64(1), (em >> np.uint64(31)) & np.uint64(1)], axis=1).astype(bool)
                assert (side_bits[:, 0] >= new_rk[:, :nA].any(axis=1)).all() and (side_bits[:, 1] >= new_rk[:, nA:].any(axis=1)).all()
                launches += int(side_bits.sum())
            else:
                unused = (em >> np.uint64(nU)) & np.uint64((1 << (8 - nU)) - 1) if nU < 8 else np.zeros_like(em)
                assert not unused.any(), "bits of slots without an aircraft"
                lf = ((em[:, None] >> np.uint64(24)) >> np.arange(nU, dtype=np.uint64)[None, :] & np.uint64(1)).astype(bool)
                new_rk = (rk_before == 0) & (rk_after == 1)
                assert (lf >= new_rk).all(), "a new rocket without its launch bit"
                launches += int(lf.sum())
            deaths += int(died.sum())
            if running == 0:
                break
        o.hl_end()
    assert deaths > 0 and launches > 0
