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
