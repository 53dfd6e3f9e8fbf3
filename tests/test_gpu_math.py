"""The shared math headers evaluated on the MI355X (hh_math_eval) against the same headers compiled for the host (the oracle's
hho_math_eval): bit for bit, operand by operand — the direct form of what the trajectory parity tests show indirectly."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(fn, a, b=None, c=None, d=None):
    import torch
    from hhmarl_2d_amd import _lib as L
    ta = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tb = torch.from_numpy(np.ascontiguousarray(b)).cuda() if b is not None else None
    o0 = torch.from_numpy(np.ascontiguousarray(c)).cuda() if c is not None else torch.zeros_like(ta)
    o1 = torch.from_numpy(np.ascontiguousarray(d)).cuda() if d is not None else torch.zeros_like(ta)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    L.check(L.lib().hh_math_eval(fn, ta.numel(), p(ta), p(tb), p(o0), p(o1), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return o0.cpu().numpy(), o1.cpu().numpy()


def _same(x, y):
    return np.array_equal(x.view(np.uint64), y.view(np.uint64)) or bool(np.all((x.view(np.uint64) == y.view(np.uint64)) | (np.isnan(x) & np.isnan(y))))


def test_device_math_equals_host_math_bit_for_bit(oracle):
    rng = np.random.default_rng(11)
    n = 400_000
    ang = np.concatenate([rng.uniform(-800, 800, n), [0.0, -0.0, 90.0, 180.0, 270.0, 360.0, 359.99999999999994, -1e-300, 45.0]])
    unit = np.concatenate([rng.uniform(-1, 1, n), [1.0, -1.0, 0.0, -0.0, 1 - 1e-16, 0.5, -0.5, 1e-300, 0.9999999999999999]])
    y, x = rng.normal(size=ang.size) * 10.0 ** rng.integers(-6, 6, ang.size), rng.normal(size=ang.size) * 10.0 ** rng.integers(-6, 6, ang.size)
    y[:8] = [0.0, -0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 3.0]
    x[:8] = [1.0, 1.0, -1.0, 0.0, 0.0, 0.0, -0.0, -0.0]
    pos = np.concatenate([rng.uniform(0, 4, n) ** 8, [0.0, 1.0, 2.0, 4.0, 1e-300, 1e300, 0.25, 1 - 2 ** -53, 1 + 2 ** -52]])   # sqrt: zero, or >= 2^-767
    cases = [
        (0, np.radians(ang), None), (1, y, x), (2, unit, None), (3, ang, None), (4, y, x),
        (5, ang, np.full(ang.size, 360.0)), (5, ang, np.full(ang.size, 359.0)), (6, ang, np.full(ang.size, 360.0)), (7, ang, np.full(ang.size, 360.0)),
        (8, unit, None), (9, np.rint(unit * 1000), np.full(unit.size, 1000.0)), (9, ang, np.full(ang.size, 180.0)),
        (11, pos, None), (12, unit * 2, None), (13, ang, np.full(ang.size, 359.0)),
    ]
    for fn, a, b in cases:
        want0, want1 = oracle.math_eval(fn, a, b if b is not None else np.zeros_like(a))
        got0, got1 = _dev(fn, a, b)
        assert _same(got0, want0), f"fn {fn}: first output differs on {int((got0.view(np.uint64) != want0.view(np.uint64)).sum())} operands"
        if fn in (0, 3):
            assert _same(got1, want1), f"fn {fn}: second output"
    # the one-turn modulo on its interval
    for m in (360.0, 359.0):
        a = np.concatenate([rng.uniform(-m, 2 * m, n), [-m, 0.0, -0.0, m, np.nextafter(2 * m, 0), -5e-324]])
        assert _same(_dev(10, a, np.full(a.size, m))[0], oracle.math_eval(10, a, np.full(a.size, m))[0])
    # the position update of the tick
    lat, lon = rng.uniform(30, 40, n), rng.uniform(30, 40, n)
    azi, s = rng.uniform(0, 360, n), rng.uniform(0, 450, n)
    g0, g1 = _dev(14, lat, lon, azi, s)
    w0, w1 = oracle.geo_move(lat, lon, azi, s)
    assert _same(g0, w0) and _same(g1, w1)
