"""include/hh_math.h against libm / mpmath: the bit-reproducible primitives must stay within
2 ulp of the correctly rounded result on the ranges this workload uses, and the modulo family
must be exact."""
import math

import mpmath as mp
import numpy as np


def ulps(x, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(x - ref) / np.spacing(np.abs(ref) + 1e-300)


def test_sincos(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-50, 50, 50000), rng.uniform(-1e-3, 1e-3, 10000)])
    s, c = oracle.math_eval(0, x)
    assert ulps(s, np.sin(x)).max() <= 2 and ulps(c, np.cos(x)).max() <= 2


def test_sincos_vs_mpmath(oracle):
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    x = rng.uniform(-7, 7, 300)
    s, c = oracle.math_eval(0, x)
    rs = np.array([float(mp.sin(mp.mpf(float(v)))) for v in x])
    rc = np.array([float(mp.cos(mp.mpf(float(v)))) for v in x])
    assert ulps(s, rs).max() <= 1 and ulps(c, rc).max() <= 1


def test_atan2_acos(oracle):
    rng = np.random.default_rng(2)
    y, x = rng.normal(size=50000), rng.normal(size=50000)
    a, _ = oracle.math_eval(1, y, x)
    assert ulps(a, np.arctan2(y, x)).max() <= 2
    y2 = y * 10.0 ** rng.uniform(-8, 0, y.size)
    a, _ = oracle.math_eval(1, y2, x)
    assert ulps(a, np.arctan2(y2, x)).max() <= 2
    v = np.concatenate([rng.uniform(-1, 1, 50000), 1 - 10.0 ** rng.uniform(-16, -1, 20000),
                        -(1 - 10.0 ** rng.uniform(-16, -1, 20000)), [1.0, -1.0, 0.0]])
    a, _ = oracle.math_eval(2, v)
    assert ulps(a, np.arccos(v)).max() <= 2


def test_degree_helpers(oracle):
    s, c = oracle.math_eval(3, np.array([0.0, 90, 180, 270, 360, -90, 45, 30]))
    assert list(s[:6]) == [0, 1, 0, -1, 0, -1] and list(c[:6]) == [1, 0, -1, 0, 1, 0]
    rng = np.random.default_rng(3)
    x = rng.uniform(-720, 720, 20000)
    s, c = oracle.math_eval(3, x)
    assert np.abs(s - np.sin(np.radians(x))).max() < 3e-15 and np.abs(c - np.cos(np.radians(x))).max() < 3e-15
    y, xx = rng.normal(size=20000), rng.normal(size=20000)
    a, _ = oracle.math_eval(4, y, xx)
    assert np.abs(a - np.degrees(np.arctan2(y, xx))).max() < 1e-13


def test_modulo_family_exact(oracle):
    rng = np.random.default_rng(4)
    a = np.concatenate([rng.uniform(-1000, 1000, 20000), [-1e-17, 360.0, -360.0, 0.0, -0.0, 720.5, 359.99999999999994]])
    for m in (360.0, 359.0):
        b = np.full(a.size, m)
        assert np.array_equal(oracle.math_eval(5, a, b)[0], np.array([u % m for u in a]))
        assert np.array_equal(oracle.math_eval(7, a, b)[0], np.fmod(a, m))
        # the one-turn form the kernels call on headings: same value for every operand, -0.0 / +0.0 included
        edge = np.array([-m, -m + 1e-13, np.nextafter(-m, 0), -1e-300, -5e-324, 5e-324, 0.0, -0.0, m, np.nextafter(m, 0), np.nextafter(m, 1e9),
                         np.nextafter(2 * m, 0), 2 * m - 1e-9])
        aa = np.concatenate([a[(a >= -m) & (a < 2 * m)], edge, rng.uniform(-m, 2 * m, 200000)])   # its precondition: -m <= x < 2m
        got = oracle.math_eval(10, aa, np.full(aa.size, m))[0]
        want = np.array([u % m for u in aa])
        assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want))
    b = np.full(a.size, 360.0)
    assert np.array_equal(oracle.math_eval(6, a, b)[0], np.array([math.remainder(u, 360.0) for u in a]))
    t = np.array([180.0, -180, 540, 900, -540])
    assert np.array_equal(oracle.math_eval(6, t, np.full(5, 360.0))[0], np.array([math.remainder(u, 360.0) for u in t]))


def test_round3_quotient_is_the_division(oracle):
    z = np.arange(-10 ** 6, 10 ** 6 + 1, dtype=np.float64)   # hh_round3 divides the rounded integer by 1000 with hh_div_known
    assert np.array_equal(oracle.math_eval(9, z, np.full(z.size, 1000.0))[0], z / 1000.0)


def test_round3(oracle):
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, 20000)
    assert np.array_equal(oracle.math_eval(8, x)[0], np.array([round(v, 3) for v in x]))


def test_keyed_rng_matches_python_mirror(oracle):
    import ref_harness as H  # pure-python mirror of include/hh_rng.h (no reference import needed)
    lib = oracle.lib()
    for seed, arena, ep, tick, unit, site, sub in [(0, 0, 1, 0, 0, 1, 0), (1234, 7, 3, 150, 4, 21, 2), (2**63 + 5, 65535, 9, 499, 6, 24, 0)]:
        want = H.u01(H.tick_key(H.arena_key(seed, arena), ep, tick), unit, site, sub)
        assert lib.hho_rng_u01(seed, arena, ep, tick, unit, site, sub) == want


def test_division_by_a_known_divisor_is_the_division(oracle):
    """hh_div_known (Markstein correction with the reciprocal taken once) must give the correctly rounded quotient:
    the kernels use it for every normalisation of the observation (x/180, x/359, x/max_speed, x/map extent)"""
    rng = np.random.default_rng(6)
    n = 2_000_000
    for c in (180.0, 359.0, 900.0, 600.0, 0.3, 0.29999999999999982, 0.30000000000000071, 0.5, 0.7):
        x = np.concatenate([rng.uniform(-400.0, 1000.0, n), rng.uniform(0.0, 1.0, n) * c, 10.0 ** rng.uniform(-12, 3, n // 4),
                            c * rng.integers(0, 4000, n // 4) / 1024.0, [0.0, -0.0, c, -c, 2 * c, 180.0, 179.99999999999997]])
        q, _ = oracle.math_eval(9, x, np.full(x.size, c))
        assert np.array_equal(q, x / c), c
