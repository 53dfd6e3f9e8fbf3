/*
 * hh_envelope.h — cheap, exactness-preserving stages of the weapon-envelope predicates, shared by the HIP kernels
 * and the CPU tests (tests/test_geodesic.py checks every stage against the Karney solution).
 */
#ifndef HH_ENVELOPE_H
#define HH_ENVELOPE_H

#include "hh_geodesic.h"
#include "hh_spec.h"

/* ac1.py:72-79,135-146: a missile leaves when range <= 111 km and int(|sdiff(hdg + 60, bearing)|) <= 60, i.e.
 * the bearing relative to the heading, beta = bearing - hdg in (-180, 180], satisfies -1 < beta < 121.
 * The observation path already holds, for every pair of aircraft, the PLANAR angle between the heading vector
 * and the (dlon, dlat) vector to the other aircraft (env_base.py:424-432 "focus", degrees, unsigned) and the
 * planar separation in degrees.  Signed with the cross product it is a bearing estimate beta_p whose distance to
 * the geodesic beta is bounded for |lat| <= 10 deg, 0.01 deg <= separation <= 0.85 deg by
 *     atan(|rho - 1| / (2 sqrt(rho))),  rho = N cos(phi_m) / M in [0.99125, 1.00674]      <= 0.2518 deg  (anisotropy)
 *   + |dlam| sin(phi_m) / 2                                                               <= 0.0738 deg  (convergence)
 *   + the error of the mid-latitude estimate against Karney (HH_GEO_EST_LONG_AZI)         <= 0.0100 deg
 *   + the 1e-10 guard in the focus quotient at separations >= 0.01 deg                    <= 0.0081 deg
 *   (+ rounding, < 1e-6 deg)                                                       total  <  0.35 deg
 * (tests/test_geodesic.py measures the difference over the domain and asserts it below 0.35).  The range clause
 * is certain there as well: a geodesic is no longer than the straight (lat, lon) segment, at most 111.33 km per
 * degree, so separation <= 0.85 deg gives range < 94.7 km.
 * The stage answers 1 (inside) / 0 (outside) only when beta_p is farther than HH_PLANAR_AZI_MARGIN = 0.5 deg from
 * both edges, else -1 and the caller decides with the estimate / Karney stages: the mask bit is the exact one. */
#define HH_PLANAR_MIN_DEG 0.01
#define HH_PLANAR_AZI_MARGIN 0.5
HH_HD int hh_missile_cone_planar(double lat1, double lon1, double lat2, double lon2, double focus_deg, double cross,
                                 double sep_deg) {
    const int dom = (hh_fabs(lat1) <= HH_GEO_EST_MAX_LAT) & (hh_fabs(lat2) <= HH_GEO_EST_MAX_LAT) & (hh_fabs(lon1) < 170.0) &
                    (hh_fabs(lon2) < 170.0) & (sep_deg >= HH_PLANAR_MIN_DEG) & (sep_deg <= HH_GEO_EST_LONG_DEG);
    /* cross = east(heading) * dlat - north(heading) * dlon > 0: the target is to the left, the bearing is smaller */
    const double beta = cross < 0.0 ? focus_deg : -focus_deg;
    /* straight-line (the kernels evaluate it on every lane behind one wave-uniform test): the three answers as selects */
    const int inside = (beta >= -1.0 + HH_PLANAR_AZI_MARGIN) & (beta <= (HH_MISSILE_HALF_DEG * 2.0 + 1.0) - HH_PLANAR_AZI_MARGIN);
    const int outside = (beta < -1.0 - HH_PLANAR_AZI_MARGIN) | (beta > (HH_MISSILE_HALF_DEG * 2.0 + 1.0) + HH_PLANAR_AZI_MARGIN);
    const int r = inside ? 1 : (outside ? 0 : -1);
    return dom ? r : -1;
}

/* Cannon cone (ac1.py:106-115,135-141): range < R km and |sdiff(heading, bearing)| <= w (5 / 3.5 deg).  Nine
 * candidate pairs in ten lie far outside that narrow cone.  With the heading as the planar vector (east, north) =
 * (he, hn) and the planar vector (dlon, dlat) to the target, the angle between the two differs from the geodesic
 * relative bearing by the same terms as above (no focus quotient here: no 1e-10 guard), < 0.26 deg for |lat| <= 10 deg
 * and |dlat|, |dlon| <= 0.06 deg.  Returns 1 only when the planar angle exceeds w + HH_PLANAR_CONE_MARGIN: the target is
 * certainly outside the cone and the exact test can be skipped; 0 means "not decided here". */
/* (he, hn) need not be the exactly rounded heading vector: the two-wave 2-vs-2 kernel passes the exact vector of the tick before rotated by the turn
 * just made, whose direction is within 1e-9 rad = 6e-8 deg of the exact one — inside the 0.04 deg the margin below leaves over the 0.26 deg bound. */
#define HH_PLANAR_CONE_MARGIN 0.3
#define HH_COS_5P3_DEG 0.9957246981845821 /* cos((10 / 2 + 0.3) deg) */
#define HH_COS_3P8_DEG 0.99780146829205   /* cos((7 / 2 + 0.3) deg) */
HH_HD int hh_cannon_cone_planar_outside(double lat1, double lon1, double lat2, double lon2, double he, double hn, int ac_type) {
    const double dx = lon2 - lon1, dy = lat2 - lat1;
    const int dom = (hh_fabs(lat1) <= HH_GEO_EST_MAX_LAT) & (hh_fabs(lat2) <= HH_GEO_EST_MAX_LAT) & (hh_fabs(lon1) < 170.0) &
                    (hh_fabs(lon2) < 170.0) & (hh_fabs(dx) <= HH_GEO_EST_SHORT_DEG) & (hh_fabs(dy) <= HH_GEO_EST_SHORT_DEG);
    const double d2 = dx * dx + dy * dy;
    const int ok = dom & !(d2 < 1e-10); /* below ~1 m the bearing of the segment is not defined well enough */
    const double dot = he * dx + hn * dy;
    const double h2 = he * he + hn * hn;
    const double cw = ac_type == 1 ? HH_COS_5P3_DEG : HH_COS_3P8_DEG;
    return ok & ((dot <= 0.0) | (dot * dot < (cw * cw) * (d2 * h2))); /* straight-line: no region for the GPU's exec mask to skip */
}

#endif /* HH_ENVELOPE_H */
