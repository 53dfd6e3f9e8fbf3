"""Geodesic layer pins (geographiclib==2.0 is un-vendored and absent: parity at that boundary is
unpinned by the reference, so it is pinned against mathematics):
  (i) Karney 2013 worked examples, (ii) independent mpmath ODE vectors
  (tests/golden/geodesic_ode.json from oracle/gen_geodesic_golden.py), (iii) round trips and
  closed forms, (iv) include/hh_geodesic.h (bit-reproducible math) == oracle/geodesic_ref.py (libm)."""
import json
import math
import os

import numpy as np

import geodesic_ref as G

HERE = os.path.dirname(os.path.abspath(__file__))


def _ode():
    with open(os.path.join(HERE, "golden", "geodesic_ode.json")) as fh:
        return json.load(fh)["records"]


def test_paper_examples_python_and_c(oracle):
    lat2, lon2 = G.direct(40.0, 0.0, 30.0, 10_000_000.0)
    assert abs(lat2 - 41.79331020506) < 1e-10 and abs(lon2 - 137.84490004377) < 1e-10
    s12, azi1 = G.inverse(-30.12345, 0.0, -30.12344, 0.00005)
    assert abs(s12 - 4.944208) < 1e-6 and abs(azi1 - 77.04353354237) < 2e-7
    la, lo = oracle.geo_direct([40.0], [0.0], [30.0], [1e7])
    assert abs(la[0] - 41.79331020506) < 1e-10 and abs(lo[0] - 137.84490004377) < 1e-10
    s, a = oracle.geo_inverse([-30.12345], [0.0], [-30.12344], [0.00005])
    assert abs(s[0] - 4.944208) < 1e-6 and abs(a[0] - 77.04353354237) < 2e-7


def test_against_mpmath_ode(oracle):
    R = _ode()
    A = lambda k: np.array([r[k] for r in R])
    for direct, inverse in ((lambda *a: np.array([G.direct(*t) for t in zip(*a)]).T,
                             lambda *a: np.array([G.inverse(*t) for t in zip(*a)]).T),
                            (oracle.geo_direct, oracle.geo_inverse)):
        la, lo = direct(A("lat1"), A("lon1"), A("azi1"), A("s12"))
        assert np.abs(la - A("lat2")).max() < 2e-14 and np.abs(lo - A("lon2")).max() < 2e-14
        s, a = inverse(A("lat1"), A("lon1"), A("lat2"), A("lon2"))
        assert np.abs(s - A("s12")).max() < 5e-9
        cross = np.abs(np.radians((a - A("azi1") + 180) % 360 - 180)) * A("s12")
        assert cross.max() < 5e-9  # metres


def test_c_matches_python_reference(oracle):
    rng = np.random.default_rng(11)
    n = 4000
    lat, lon = rng.uniform(4.9, 5.6, n), rng.uniform(6.9, 7.6, n)
    az, s = rng.uniform(0, 360, n), rng.uniform(0, 1100, n)
    az[:500] = rng.integers(0, 360, 500)
    la, lo = oracle.geo_direct(lat, lon, az, s)
    ref = np.array([G.direct(*t) for t in zip(lat, lon, az, s)])
    assert np.abs(la - ref[:, 0]).max() < 5e-14 and np.abs(lo - ref[:, 1]).max() < 5e-14
    lat2, lon2 = rng.uniform(4.9, 5.6, n), rng.uniform(6.9, 7.6, n)
    lat2[:1000] = lat[:1000] + rng.normal(0, 0.01, 1000)
    lon2[:1000] = lon[:1000] + rng.normal(0, 0.01, 1000)
    lon2[1000:1050] = lon[1000:1050]           # meridional
    lat2[1050:1060] = lat[1050:1060]           # same parallel
    lat2[1060:1065], lon2[1060:1065] = lat[1060:1065], lon[1060:1065]  # coincident
    s12, a1 = oracle.geo_inverse(lat, lon, lat2, lon2)
    ref = np.array([G.inverse(*t) for t in zip(lat, lon, lat2, lon2)])
    assert np.abs(s12 - ref[:, 0]).max() < 1e-8
    assert (np.abs(np.radians((a1 - ref[:, 1] + 180) % 360 - 180)) * ref[:, 0]).max() < 1e-8


def test_round_trip_and_closed_forms(oracle):
    rng = np.random.default_rng(12)
    n = 3000
    lat, lon = rng.uniform(5, 5.5, n), rng.uniform(7, 7.5, n)
    az, s = rng.uniform(0, 360, n), rng.uniform(1, 80000, n)
    la, lo = oracle.geo_direct(lat, lon, az, s)
    s2, a2 = oracle.geo_inverse(lat, lon, la, lo)
    assert np.abs(s2 - s).max() < 1e-8
    assert (np.abs(np.radians((a2 - az + 180) % 360 - 180)) * s).max() < 1e-7
    # equator: s = a * dlambda
    s_eq, a_eq = oracle.geo_inverse([0.0], [7.1], [0.0], [7.2])
    assert abs(s_eq[0] - 6378137.0 * math.radians(0.1)) < 1e-8 and a_eq[0] == 90.0
    # meridian: azimuth exactly 0 / 180, symmetric distance
    s_n, a_n = oracle.geo_inverse([5.1], [7.1], [5.2], [7.1])
    s_s, a_s = oracle.geo_inverse([5.2], [7.1], [5.1], [7.1])
    assert a_n[0] == 0.0 and a_s[0] == 180.0 and s_n[0] == s_s[0]
    # zero-length step leaves the point in place (speed 0 opponents, cmano_simulator.py:67)
    la, lo = oracle.geo_direct([5.2], [7.2], [123.0], [0.0])
    assert abs(la[0] - 5.2) < 1e-14 and abs(lo[0] - 7.2) < 1e-14


# ---------------------------------------------------------------- the cheap stages the kernels put in front of Karney
def _pairs(rng, n, max_sep, min_sep=2e-5, lat_lim=10.0):
    lat1 = rng.uniform(-lat_lim, lat_lim, n)
    lon1 = rng.uniform(-160, 160, n)
    sep = np.exp(rng.uniform(np.log(min_sep), np.log(max_sep), n))
    th = rng.uniform(0, 2 * np.pi, n)
    lat2 = np.clip(lat1 + sep * np.cos(th), -lat_lim, lat_lim)
    lon2 = lon1 + sep * np.sin(th)
    return lat1, lon1, lat2, lon2, np.degrees(th)


def test_short_step_direct_against_karney(oracle):
    """hh_geo_move (4th-order series for one tick's displacement, include/hh_geodesic.h) against the Karney Direct:
    the position update of every aircraft and rocket (cmano_simulator.py:65-72)"""
    rng = np.random.default_rng(5)
    n = 400_000
    lat, lon = rng.uniform(-60, 60, n), rng.uniform(-160, 160, n)
    lat[: n // 2] = rng.uniform(4.5, 6.0, n // 2)   # the arena
    lon[: n // 2] = rng.uniform(6.5, 8.0, n // 2)
    az = rng.uniform(0, 360, n)
    az[::7] = np.floor(az[::7])
    s = rng.uniform(0.0, 1100.0, n)                 # <= 2000 kn * 0.514444 m/s * 1 s
    s[::11] = 0.0
    la_s, lo_s = oracle.geo_move(lat, lon, az, s)
    la_k, lo_k = oracle.geo_direct(lat, lon, az, s)
    assert np.abs(la_s - la_k).max() < 2e-13 and np.abs(lo_s - lo_k).max() < 4e-13   # degrees: < 5e-8 m
    # zero displacement leaves the point exactly in place
    z = s == 0.0
    assert np.array_equal(la_s[z], lat[z]) and np.array_equal(lo_s[z], lon[z])


def test_inverse_estimate_error_bounds(oracle):
    """the mid-latitude estimate may only decide an envelope test when it is farther from the threshold than
    HH_GEO_EST_* (hh_geodesic.h); measured error must stay >= 10x below those bounds over the whole domain"""
    rng = np.random.default_rng(6)
    n = 600_000
    for max_sep, rel, abs_m, azi in ((0.06, 1e-6, 1e-3, 1e-4), (0.85, 0.0, 10.0, 1e-2)):
        lat1, lon1, lat2, lon2, _ = _pairs(rng, n, max_sep, min_sep=2e-5)
        # keep |dlat|, |dlon| inside the stage's box
        ok = (np.abs(lat2 - lat1) <= max_sep) & (np.abs(lon2 - lon1) <= max_sep)
        lat1, lon1, lat2, lon2 = lat1[ok], lon1[ok], lat2[ok], lon2[ok]
        s_e, a_e = oracle.geo_inverse_estimate(lat1, lon1, lat2, lon2)
        s_k, a_k = oracle.geo_inverse(lat1, lon1, lat2, lon2)
        far = s_k > 1.0                              # HH_GEO_EST_MIN_M: closer pairs are never decided by the estimate
        ds = np.abs(s_e - s_k)[far]
        da = np.abs((a_e - a_k + 180.0) % 360.0 - 180.0)[far]
        assert (ds <= 0.1 * (rel * s_k[far] + abs_m)).all(), (max_sep, ds.max())
        assert da.max() <= 0.1 * azi, (max_sep, da.max())


def test_planar_missile_cone_stage(oracle):
    """hh_missile_cone_planar (hh_envelope.h): whenever the planar stage answers, the answer is the exact launch
    predicate (ac1.py:72-79,135-146); the planar/geodesic bearing gap stays below the 0.35 deg the margin assumes"""
    rng = np.random.default_rng(7)
    n = 1_500_000
    lat1, lon1, lat2, lon2, brg = _pairs(rng, n, 0.85, min_sep=0.01)
    edge = rng.choice([-1.0, 121.0], n) + rng.uniform(-2, 2, n)
    hdg = np.where(rng.random(n) < 0.5, (brg - edge) % 360.0, rng.uniform(0, 360, n))
    hdg = np.where(rng.random(n) < 0.3, np.floor(hdg), hdg)     # MultiDiscrete headings are whole degrees
    pre, exact, bp, bg = oracle.missile_cone_planar(lat1, lon1, hdg, lat2, lon2)
    gap = np.abs((bp - bg + 180.0) % 360.0 - 180.0)
    assert gap.max() < 0.35
    decided = pre >= 0
    assert (pre[decided] == exact[decided]).all()
    assert decided.mean() > 0.8
    # inside the arena (lat 5..5.3) nearly every launch is decided here
    la1, lo1 = rng.uniform(5.0, 5.3, n), rng.uniform(7.0, 7.3, n)
    la2, lo2 = rng.uniform(5.0, 5.3, n), rng.uniform(7.0, 7.3, n)
    hd = np.floor(rng.uniform(0, 360, n))
    pre, exact, bp, bg = oracle.missile_cone_planar(la1, lo1, hd, la2, lo2)
    decided = pre >= 0
    assert (pre[decided] == exact[decided]).all() and decided.mean() > 0.97
    # outside the stage's domain it never answers
    pre, _, _, _ = oracle.missile_cone_planar([5.0, 5.0, 40.0], [7.0, 7.0, 7.0], [0.0, 0.0, 0.0], [5.0, 6.5, 40.1], [7.001, 7.0, 7.0])
    assert (pre == -1).all()


def test_planar_cannon_cone_stage(oracle):
    """hh_cannon_cone_planar_outside: 'certainly outside the cone' must imply the exact predicate is false"""
    rng = np.random.default_rng(8)
    n = 1_000_000
    for t, w in ((1, 5.0), (2, 3.5)):
        lat1, lon1, lat2, lon2, brg = _pairs(rng, n, 0.08, min_sep=2e-6)
        edge = rng.choice([-1.0, 1.0], n) * (w + rng.uniform(-0.6, 0.6, n))
        hdg = np.where(rng.random(n) < 0.7, (brg - edge) % 360.0, rng.uniform(0, 360, n))
        hdg = np.where(rng.random(n) < 0.3, np.floor(hdg), hdg)
        out, exact = oracle.cannon_cone_planar(t, lat1, lon1, hdg, lat2, lon2)
        assert not ((out == 1) & (exact == 1)).any()
        assert out.mean() > 0.3
        # coincident points are left to the exact stage
        out0, _ = oracle.cannon_cone_planar(t, [5.1], [7.1], [33.0], [5.1], [7.1])
        assert out0[0] == 0
