"""
TEST INFRASTRUCTURE — generates tests/golden/geodesic_ode.json.

Independent oracle for the geodesic layer: 30-digit mpmath integration of the geodesic
equations on the WGS84 ellipsoid,
    dphi/ds = cos(alpha) / M(phi),  dlam/ds = sin(alpha) / (N(phi) cos(phi)),
    dalpha/ds = sin(alpha) tan(phi) / N(phi),
with arclength non-dimensionalised by the semi-major axis (integrating in metres does not
terminate in reasonable time).  Shares no code and no series with oracle/geodesic_ref.py or
include/hh_geodesic.h.  Each record is one geodesic segment: (lat1, lon1, azi1, s12) ->
(lat2, lon2), which pins Direct forwards and Inverse backwards.

Run:  python oracle/gen_geodesic_golden.py   (about a minute; needs mpmath)
"""
import json
import os
import random

import mpmath as mp

mp.mp.dps = 30
A = mp.mpf(6378137)
F = 1 / mp.mpf("298.257223563")
E2 = F * (2 - F)


def ode_direct(lat1, lon1, azi1, s12):
    phi0 = mp.radians(mp.mpf(lat1))
    lam0 = mp.radians(mp.mpf(lon1))
    alp0 = mp.radians(mp.mpf(azi1))
    S = mp.mpf(s12) / A  # non-dimensional arclength

    def rhs(t, y):
        phi, lam, alp = y
        w2 = 1 - E2 * mp.sin(phi) ** 2
        w = mp.sqrt(w2)
        n = 1 / w                     # N/a
        m = (1 - E2) / (w2 * w)       # M/a
        return [mp.cos(alp) / m, mp.sin(alp) / (n * mp.cos(phi)), mp.sin(alp) * mp.tan(phi) / n]

    if S == 0:
        return float(lat1), float(lon1)
    sol = mp.odefun(rhs, 0, [phi0, lam0, alp0], tol=mp.mpf(10) ** -26, degree=None)
    phi, lam, _ = sol(S)
    return float(mp.degrees(phi)), float(mp.degrees(lam))


def main():
    rng = random.Random(20130101)
    recs = []
    # per-tick steps (aircraft 0..463 m, rockets up to 1029 m) and arena-scale separations
    dists = [0.5, 25.0, 51.4444, 180.05, 463.0, 1028.888] + [rng.uniform(1.0, 1030.0) for _ in range(10)] \
        + [1999.0, 4500.0, 8100.0] + [rng.uniform(1e3, 8e4) for _ in range(11)]
    for s in dists:
        lat = rng.uniform(5.0, 5.5)
        lon = rng.uniform(7.0, 7.5)
        azi = rng.choice([0.0, 90.0, 180.0, 270.0, 45.0]) if rng.random() < 0.2 else rng.uniform(0.0, 360.0)
        lat2, lon2 = ode_direct(lat, lon, azi, s)
        recs.append({"lat1": lat, "lon1": lon, "azi1": azi, "s12": s, "lat2": lat2, "lon2": lon2})
        print(recs[-1])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "geodesic_ode.json")
    with open(out, "w") as fh:
        json.dump({"generator": "oracle/gen_geodesic_golden.py", "dps": 30, "records": recs}, fh, indent=1)


if __name__ == "__main__":
    main()
