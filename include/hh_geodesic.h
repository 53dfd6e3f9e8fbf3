/*
 * hh_geodesic.h — WGS84 geodesic Direct / Inverse in FP64, host + device.
 *
 * Replaces (reference call sites): warsim/utils/geodesics.py:12-14 geodetic_distance_km,
 * :17-19 geodetic_bearing_deg, :22-24 geodetic_direct — i.e. geographiclib 2.0's
 * Geodesic.WGS84.Inverse / .Direct, which is an un-vendored third-party dependency
 * (README.md:22) and therefore restated from the published algorithm:
 *   C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87 (2013) 43-55, order-6 series.
 *
 * Same algorithm as oracle/geodesic_ref.py (which runs on libm and is pinned against an
 * independent mpmath ODE integration); this file runs on the bit-reproducible hh_math.h
 * primitives so that the plain-C oracle and the gfx950 kernels produce identical bits.
 * Compile both sides with -ffp-contract=off.
 *
 * Domain handled: |lat| < 90, separations far from antipodal (the nearly-antipodal start and
 * the bisection fallback of the full algorithm are unreachable for arena-scale geometry and
 * are not restated).  Meridional, equatorial and coincident cases are handled.
 */
#ifndef HH_GEODESIC_H
#define HH_GEODESIC_H

#include "hh_math.h"

#define HH_GEO_A 6378137.0
#define HH_GEO_F 0.0033528106647474805
#define HH_GEO_F1 0.9966471893352525
#define HH_GEO_E2 0.0066943799901413165
#define HH_GEO_EP2 0.006739496742276434
#define HH_GEO_N 0.0016792203863837047
#define HH_GEO_B 6356752.314245179
#define HH_GEO_ETOL2 3.6424611488788524e-08
#define HH_GEO_TINY 1.4916681462400413e-154
#define HH_GEO_TOL0 2.220446049250313e-16
#define HH_GEO_MAXIT 20

/* sum_{l=1..6} c[l] sin(2 l x) by Clenshaw summation (c[0] unused) */
HH_HD double hh_geo_clenshaw6(double sinx, double cosx, const double *c) {
    double ar = 2.0 * (cosx - sinx) * (cosx + sinx);
    double y0 = 0.0, y1 = 0.0;
    y1 = ar * y0 - y1 + c[6];
    y0 = ar * y1 - y0 + c[5];
    y1 = ar * y0 - y1 + c[4];
    y0 = ar * y1 - y0 + c[3];
    y1 = ar * y0 - y1 + c[2];
    y0 = ar * y1 - y0 + c[1];
    return 2.0 * sinx * cosx * y0;
}

/* sum_{l=1..5} c[l] sin(2 l x) */
HH_HD double hh_geo_clenshaw5(double sinx, double cosx, const double *c) {
    double ar = 2.0 * (cosx - sinx) * (cosx + sinx);
    double y0 = c[5], y1 = 0.0;
    y1 = ar * y0 - y1 + c[4];
    y0 = ar * y1 - y0 + c[3];
    y1 = ar * y0 - y1 + c[2];
    y0 = ar * y1 - y0 + c[1];
    return 2.0 * sinx * cosx * y0;
}

HH_HD double hh_geo_A1m1f(double eps) {
    double e2 = eps * eps;
    double t = e2 * (e2 * (e2 + 4.0) + 64.0) / 256.0;
    return (t + eps) / (1.0 - eps);
}

HH_HD void hh_geo_C1f(double eps, double *c) {
    double e2 = eps * eps, d = eps;
    c[1] = d * ((6.0 - e2) * e2 - 16.0) / 32.0;
    d *= eps;
    c[2] = d * ((64.0 - 9.0 * e2) * e2 - 128.0) / 2048.0;
    d *= eps;
    c[3] = d * (9.0 * e2 - 16.0) / 768.0;
    d *= eps;
    c[4] = d * (3.0 * e2 - 5.0) / 512.0;
    d *= eps;
    c[5] = -7.0 * d / 1280.0;
    d *= eps;
    c[6] = -7.0 * d / 2048.0;
}

HH_HD void hh_geo_C1pf(double eps, double *c) {
    double e2 = eps * eps, d = eps;
    c[1] = d * (e2 * (205.0 * e2 - 432.0) + 768.0) / 1536.0;
    d *= eps;
    c[2] = d * (e2 * (4005.0 * e2 - 4736.0) + 3840.0) / 12288.0;
    d *= eps;
    c[3] = d * (116.0 - 225.0 * e2) / 384.0;
    d *= eps;
    c[4] = d * (2695.0 - 7173.0 * e2) / 7680.0;
    d *= eps;
    c[5] = 3467.0 * d / 7680.0;
    d *= eps;
    c[6] = 38081.0 * d / 61440.0;
}

HH_HD double hh_geo_A2m1f(double eps) {
    double e2 = eps * eps;
    double t = e2 * (e2 * (-11.0 * e2 - 28.0) - 192.0) / 256.0;
    return (t - eps) / (1.0 + eps);
}

HH_HD void hh_geo_C2f(double eps, double *c) {
    double e2 = eps * eps, d = eps;
    c[1] = d * (e2 * (e2 + 2.0) + 16.0) / 32.0;
    d *= eps;
    c[2] = d * (e2 * (35.0 * e2 + 64.0) + 384.0) / 2048.0;
    d *= eps;
    c[3] = d * (15.0 * e2 + 80.0) / 768.0;
    d *= eps;
    c[4] = d * (7.0 * e2 + 35.0) / 512.0;
    d *= eps;
    c[5] = 63.0 * d / 1280.0;
    d *= eps;
    c[6] = 77.0 * d / 2048.0;
}

/* A3 and C3 with the WGS84 third flattening n folded into the coefficients (Karney eqs 24-25) */
HH_HD double hh_geo_A3f(double eps) {
    double v = -0.0234375;
    v = v * eps + -0.046927475637074494;
    v = v * eps + -0.06281503005876607;
    v = v * eps + -0.2502088451303832;
    v = v * eps + -0.49916038980680816;
    v = v * eps + 1.0;
    return v;
}

HH_HD void hh_geo_C3f(double eps, double *c) {
    double mult = eps, v;
    v = 0.0234375;
    v = v * eps + 0.03908873781853724;
    v = v * eps + 0.04695366939653196;
    v = v * eps + 0.12499964752736174;
    v = v * eps + 0.24958019490340408;
    c[1] = mult * v;
    mult *= eps;
    v = 0.01953125;
    v = v * eps + 0.02345061890926862;
    v = v * eps + 0.046822392185686165;
    v = v * eps + 0.062342661206936094;
    c[2] = mult * v;
    mult *= eps;
    v = 0.013671875;
    v = v * eps + 0.023393770302437927;
    v = v * eps + 0.025963026642854565;
    c[3] = mult * v;
    mult *= eps;
    v = 0.013671875;
    v = v * eps + 0.01362595881755982;
    c[4] = mult * v;
    mult *= eps;
    c[5] = mult * 0.008203125;
}

HH_HD void hh_geo_norm(double *x, double *y) {
    double r = hh_hypot(*x, *y);
    *x = *x / r;
    *y = *y / r;
}

/* ------------------------------------------------------------------ Direct */
/* geodesics.py:22-24: Direct(lat, lon, heading, distance) -> lat2, lon2 */
HH_HD void hh_geo_direct(double lat1, double lon1, double azi1, double s12, double *lat2, double *lon2) {
    double C1a[7], C1pa[7], C3a[6];
    double salp1, calp1, sbet1, cbet1;
    azi1 = hh_ang_normalize(azi1);
    hh_sincosd(hh_ang_round(azi1), &salp1, &calp1);
    hh_sincosd(hh_ang_round(lat1), &sbet1, &cbet1);
    sbet1 *= HH_GEO_F1;
    hh_geo_norm(&sbet1, &cbet1);
    cbet1 = hh_max(HH_GEO_TINY, cbet1);
    double salp0 = salp1 * cbet1;
    double calp0 = hh_hypot(calp1, salp1 * sbet1);
    double ssig1 = sbet1;
    double somg1 = salp0 * sbet1;
    double csig1 = (sbet1 != 0.0 || calp1 != 0.0) ? cbet1 * calp1 : 1.0;
    double comg1 = csig1;
    hh_geo_norm(&ssig1, &csig1);
    double k2 = calp0 * calp0 * HH_GEO_EP2;
    double eps = k2 / (2.0 * (1.0 + hh_sqrt(1.0 + k2)) + k2);
    double A1m1 = hh_geo_A1m1f(eps);
    hh_geo_C1f(eps, C1a);
    double B11 = hh_geo_clenshaw6(ssig1, csig1, C1a);
    double s, c;
    hh_sincos(B11, &s, &c);
    double stau1 = ssig1 * c + csig1 * s;
    double ctau1 = csig1 * c - ssig1 * s;
    hh_geo_C1pf(eps, C1pa);
    double A3c = -HH_GEO_F * salp0 * hh_geo_A3f(eps);
    hh_geo_C3f(eps, C3a);
    double B31 = hh_geo_clenshaw5(ssig1, csig1, C3a);
    double tau12 = s12 / (HH_GEO_B * (1.0 + A1m1));
    hh_sincos(tau12, &s, &c);
    double B12 = -hh_geo_clenshaw6(stau1 * c + ctau1 * s, ctau1 * c - stau1 * s, C1pa);
    double sig12 = tau12 - (B12 - B11);
    double ssig12, csig12;
    hh_sincos(sig12, &ssig12, &csig12);
    double ssig2 = ssig1 * csig12 + csig1 * ssig12;
    double csig2 = csig1 * csig12 - ssig1 * ssig12;
    double sbet2 = calp0 * ssig2;
    double cbet2 = hh_hypot(salp0, calp0 * csig2);
    if (cbet2 == 0.0) cbet2 = csig2 = HH_GEO_TINY;
    double somg2 = salp0 * ssig2;
    double comg2 = csig2;
    double omg12 = hh_atan2(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
    double lam12 = omg12 + A3c * (sig12 + (hh_geo_clenshaw5(ssig2, csig2, C3a) - B31));
    double lon12 = lam12 * HH_RAD2DEG;
    *lon2 = hh_ang_normalize(hh_ang_normalize(lon1) + hh_ang_normalize(lon12));
    *lat2 = hh_atan2d(sbet2, HH_GEO_F1 * cbet2);
}


/* ------------------------------------------------------------------ short-step Direct */
/* Per-tick position updates move at most 1.03 km (2000 kn rocket): t = s/a <= 1.7e-4.  For such steps the
 * geodesic equations on the ellipsoid,
 *     dphi/dt = cos(alp) W^3 / (1-e^2),  dlam/dt = sin(alp) W / cos(phi),  dalp/dt = sin(alp) tan(phi) W,
 *     W = sqrt(1 - e^2 sin^2 phi),
 * are solved by their TAYLOR SERIES in t to 3rd order (rounds 1-4 carried the 4th: in the domain below its term is at most one unit in the last
 * place of the result — measured: 8.9e-16 deg against the 4th-order form over 2e6 arena-scale samples, the same 4.4e-15 / 1.7e-14 deg against
 * Karney — for 32 of the function's ~150 operations on the tick's critical path).  The coefficients come from the standard
 * power-series recurrences of the auxiliary functions S = sin phi, C = cos phi, Sa = sin alp, Ca = cos alp,
 * U = 1 - e^2 S^2, W = sqrt(U), Q = W/C, P = W U/(1-e^2)   (S' = C phi', C' = -S phi', W_k from W^2 = U,
 * Q_k from Q C = W, products by Cauchy sums), all evaluated at the start point only.  The truncation
 * error is ~ t^4 |f4| < 1e-15 rad at the 1.1 km domain limit at arena latitudes (a few 1e-14 deg at 60 deg: tests/test_geodesic.py) — at double rounding — and the
 * dependency chain is ~30 operations deep instead of the ~160 of a Runge-Kutta step or the several
 * hundred of the general series solution: this is the per-tick latency that matters at one wave per
 * SIMD.  Cost: 2 division-free sincosd, 1 sqrt, 2 divisions, ~120 multiply-adds.
 * tests/test_geodesic.py pins it to hh_geo_direct (Karney) and to the mpmath ODE vectors at <= 2e-14 deg.
 * Outside its domain (s > 1.1 km: beyond one tick of the fastest unit, a 2000 kn rocket; or |lat| > 70) callers fall back to hh_geo_direct. */
#define HH_GEO_SHORT_MAX_M 1100.0
#define HH_GEO_SHORT_MAX_LAT 70.0

/* The terms of the series that depend on the START LATITUDE only (a sincosd, a square root, two divisions: the long dependent chains of the
 * function), split from the rest: hh_geo_direct_short = these terms + hh_geo_direct_short_core, the same operations on the same operands.  (Round 5
 * tried computing them a tick ahead on the 2-vs-2 kernel's output wave: +0.6 %, and the room in that wave's window went to the opponents' script.) */
typedef struct { double S0, C0, U0, W0, iC, hW; } hh_geo_lat_terms;
HH_HD hh_geo_lat_terms hh_geo_short_lat_terms(double lat1) {
    hh_geo_lat_terms t;
    hh_sincosd_small(lat1, &t.S0, &t.C0);
    t.U0 = hh_fma(-HH_GEO_E2 * t.S0, t.S0, 1.0);
    t.W0 = hh_sqrt(t.U0);
    t.iC = 1.0 / t.C0;
    t.hW = 0.5 / t.W0;
    return t;
}
HH_HD void hh_geo_direct_short_core(double lat1, double lon1, double azi1, double s12, hh_geo_lat_terms lt, double *lat2, double *lon2) {
    const double E2 = HH_GEO_E2, K1 = 1.0 / (1.0 - HH_GEO_E2), TH = 1.0 / 3.0, M2E2 = -2.0 * HH_GEO_E2;
    const double S0 = lt.S0, C0 = lt.C0, U0 = lt.U0, W0 = lt.W0, iC = lt.iC, hW = lt.hW;
    double Sa0, Ca0;
    hh_sincosd_small(azi1, &Sa0, &Ca0);
    const double h = s12 * (1.0 / HH_GEO_A);
    const double Q0 = W0 * iC, P0 = W0 * U0 * K1;
    /* every Cauchy sum below is one multiply followed by fused multiply-adds (hh_fma on both sides of the parity) */
    /* order 1 */
    const double f1 = Ca0 * P0;
    const double SS0 = Sa0 * S0;
    const double a1 = SS0 * Q0;
    const double l1 = Sa0 * Q0;
    const double S1 = C0 * f1, C1 = -S0 * f1, Sa1 = Ca0 * a1, Ca1 = -Sa0 * a1;
    const double U1 = (M2E2 * S0) * S1;
    const double W1 = U1 * hW;
    const double Q1 = hh_fma(-C1, Q0, W1) * iC;
    const double P1 = hh_fma(W0, U1, W1 * U0) * K1;
    /* order 2: g2 = 2 f2, h2 = 2 a2 */
    const double g2 = hh_fma(Ca0, P1, Ca1 * P0);
    const double f2 = 0.5 * g2;
    const double SS1 = hh_fma(Sa0, S1, Sa1 * S0);
    const double h2 = hh_fma(SS0, Q1, SS1 * Q0);
    const double l2 = 0.5 * hh_fma(Sa0, Q1, Sa1 * Q0);
    const double S2 = 0.5 * hh_fma(C0, g2, C1 * f1), C2 = -0.5 * hh_fma(S0, g2, S1 * f1);
    const double Sa2 = 0.5 * hh_fma(Ca0, h2, Ca1 * a1), Ca2 = -0.5 * hh_fma(Sa0, h2, Sa1 * a1);
    const double U2 = -E2 * hh_fma(2.0 * S0, S2, S1 * S1);
    const double W2 = hh_fma(-W1, W1, U2) * hW;
    const double Q2 = hh_fma(-C2, Q0, hh_fma(-C1, Q1, W2)) * iC;
    const double P2 = hh_fma(W0, U2, hh_fma(W1, U1, W2 * U0)) * K1;
    /* order 3: g3 = 3 f3 (the azimuth's third coefficient feeds the fourth order only: not needed) */
    const double g3 = hh_fma(Ca0, P2, hh_fma(Ca1, P1, Ca2 * P0));
    const double f3 = TH * g3;
    const double l3 = TH * hh_fma(Sa0, Q2, hh_fma(Sa1, Q1, Sa2 * Q0));
    const double dphi = h * hh_fma(h, hh_fma(h, f3, f2), f1);
    const double dlam = h * hh_fma(h, hh_fma(h, l3, l2), l1);
    *lat2 = hh_fma(dphi, HH_RAD2DEG, lat1);
    *lon2 = hh_fma(dlam, HH_RAD2DEG, lon1);
}
HH_HD void hh_geo_direct_short(double lat1, double lon1, double azi1, double s12, double *lat2, double *lon2) {
    hh_geo_direct_short_core(lat1, lon1, azi1, s12, hh_geo_short_lat_terms(lat1), lat2, lon2);
}

/* position update used by the simulator tick (cmano_simulator.py:65-72) */
HH_HD void hh_geo_move(double lat1, double lon1, double azi1, double s12, double *lat2, double *lon2) {
    if (s12 <= HH_GEO_SHORT_MAX_M && hh_fabs(lat1) <= HH_GEO_SHORT_MAX_LAT && hh_fabs(lon1) < 170.0 && hh_fabs(azi1) <= 360.0)
        hh_geo_direct_short(lat1, lon1, azi1, s12, lat2, lon2);
    else
        hh_geo_direct(lat1, lon1, azi1, s12, lat2, lon2);
}


/* ------------------------------------------------------------------ filter estimate for Inverse */
/* Mid-latitude (Gauss) estimate of range and initial bearing:
 *     x = N(phi_m) cos(phi_m) dlam,  y = M(phi_m) dphi,  s ~ hypot(x, y),
 *     azi1 ~ atan2(x, y) - dlam sin(phi_m) / 2          (half the meridian convergence).
 * It is NOT used for results.  The kernels use it as the cheap stage of a filtered exact predicate:
 * a weapon-envelope test (range < R, |bearing - heading| <= w) is decided from the estimate only
 * when the estimate is farther from the threshold than the error bound below; otherwise the
 * test is redone with hh_geo_inverse.  Either way the mask bit equals the exact predicate.
 * Error bounds (tests/test_geodesic.py measures them against hh_geo_inverse over the domain
 * |lat| <= 10 deg and asserts a >= 10x margin to the bounds used here):
 *     |dlat|,|dlon| <= 0.06 deg (<= 9.4 km), s > 1 m:  |ds| <= 2.3e-4 m, |dazi| <= 3.9e-6 deg -> bounds 1e-6*s + 1e-3 m, 1e-4 deg
 *     |dlat|,|dlon| <= 0.85 deg (<= 133 km):           |ds| <= 0.65 m,   |dazi| <= 7.8e-4 deg -> bounds 10 m, 1e-2 deg
 * Separations below 1 m are always treated as undecided (the bearing of a near-zero vector). */
#define HH_GEO_EST_MAX_LAT 10.0
#define HH_GEO_EST_SHORT_DEG 0.06
#define HH_GEO_EST_LONG_DEG 0.85
#define HH_GEO_EST_SHORT_REL 1e-6
#define HH_GEO_EST_SHORT_ABS_M 1e-3
#define HH_GEO_EST_SHORT_AZI 1e-4
#define HH_GEO_EST_LONG_ABS_M 10.0
#define HH_GEO_EST_LONG_AZI 1e-2
#define HH_GEO_EST_MIN_M 1.0

HH_HD void hh_geo_inverse_estimate(double lat1, double lon1, double lat2, double lon2, double *s12, double *azi1) {
    double pm = (0.5 * (lat1 + lat2)) * HH_DEG2RAD;
    double dp = (lat2 - lat1) * HH_DEG2RAD, dl = (lon2 - lon1) * HH_DEG2RAD;
    double sm, cm;
    hh_sincos(pm, &sm, &cm);
    double W2 = 1.0 - HH_GEO_E2 * sm * sm;
    double W = hh_sqrt(W2);
    double N = HH_GEO_A / W;
    double M = N * (1.0 - HH_GEO_E2) / W2;
    double x = N * cm * dl, y = M * dp;
    *s12 = hh_sqrt(x * x + y * y);
    double a = (hh_atan2(x, y) - 0.5 * dl * sm) * HH_RAD2DEG;
    if (a < 0.0) a += 360.0;
    if (a >= 360.0) a -= 360.0;
    *azi1 = a; /* [0, 360) */
}

/* ------------------------------------------------------------------ Inverse */
HH_HD void hh_geo_lengths(double eps, double sig12, double ssig1, double csig1, double dn1, double ssig2,
                          double csig2, double dn2, double *s12b, double *m12b) {
    double C1a[7], C2a[7];
    double A1 = hh_geo_A1m1f(eps);
    hh_geo_C1f(eps, C1a);
    double A2 = hh_geo_A2m1f(eps);
    hh_geo_C2f(eps, C2a);
    double m0x = A1 - A2;
    A2 = 1.0 + A2;
    A1 = 1.0 + A1;
    double B1 = hh_geo_clenshaw6(ssig2, csig2, C1a) - hh_geo_clenshaw6(ssig1, csig1, C1a);
    *s12b = A1 * (sig12 + B1);
    double B2 = hh_geo_clenshaw6(ssig2, csig2, C2a) - hh_geo_clenshaw6(ssig1, csig1, C2a);
    double J12 = m0x * sig12 + (A1 * B1 - A2 * B2);
    *m12b = dn2 * (csig1 * ssig2) - dn1 * (ssig1 * csig2) - csig1 * csig2 * J12;
}

typedef struct {
    double lam12, salp2, calp2, sig12, ssig1, csig1, ssig2, csig2, eps, dlam12;
} hh_geo_l12;

HH_HD void hh_geo_lambda12(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2,
                           double salp1, double calp1, double slam120, double clam120, hh_geo_l12 *o) {
    double C3a[6];
    if (sbet1 == 0.0 && calp1 == 0.0) calp1 = -HH_GEO_TINY;
    double salp0 = salp1 * cbet1;
    double calp0 = hh_hypot(calp1, salp1 * sbet1);
    double ssig1 = sbet1, somg1 = salp0 * sbet1;
    double csig1 = calp1 * cbet1, comg1 = csig1;
    hh_geo_norm(&ssig1, &csig1);
    double salp2 = cbet2 != cbet1 ? salp0 / cbet2 : salp1;
    double calp2;
    if (cbet2 != cbet1 || hh_fabs(sbet2) != -sbet1) {
        double t = cbet1 < -sbet1 ? (cbet2 - cbet1) * (cbet1 + cbet2) : (sbet1 - sbet2) * (sbet1 + sbet2);
        double cc = calp1 * cbet1;
        calp2 = hh_sqrt(cc * cc + t) / cbet2;
    } else {
        calp2 = hh_fabs(calp1);
    }
    double ssig2 = sbet2, somg2 = salp0 * sbet2;
    double csig2 = calp2 * cbet2, comg2 = csig2;
    hh_geo_norm(&ssig2, &csig2);
    double sig12 = hh_atan2(hh_max(0.0, csig1 * ssig2 - ssig1 * csig2) + 0.0, csig1 * csig2 + ssig1 * ssig2);
    double somg12 = hh_max(0.0, comg1 * somg2 - somg1 * comg2) + 0.0;
    double comg12 = comg1 * comg2 + somg1 * somg2;
    double eta = hh_atan2(somg12 * clam120 - comg12 * slam120, comg12 * clam120 + somg12 * slam120);
    double k2 = calp0 * calp0 * HH_GEO_EP2;
    double eps = k2 / (2.0 * (1.0 + hh_sqrt(1.0 + k2)) + k2);
    hh_geo_C3f(eps, C3a);
    double B312 = hh_geo_clenshaw5(ssig2, csig2, C3a) - hh_geo_clenshaw5(ssig1, csig1, C3a);
    double domg12 = -HH_GEO_F * hh_geo_A3f(eps) * salp0 * (sig12 + B312);
    o->lam12 = eta + domg12;
    if (calp2 == 0.0) {
        o->dlam12 = -2.0 * HH_GEO_F1 * dn1 / sbet1;
    } else {
        double s12b, m12b;
        hh_geo_lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12b, &m12b);
        o->dlam12 = m12b * HH_GEO_F1 / (calp2 * cbet2);
    }
    o->salp2 = salp2; o->calp2 = calp2; o->sig12 = sig12;
    o->ssig1 = ssig1; o->csig1 = csig1; o->ssig2 = ssig2; o->csig2 = csig2; o->eps = eps;
}

/* geodesics.py:12-19: Inverse(lat1, lon1, lat2, lon2) -> s12 [m], azi1 [deg in [-180,180]].
 * The reference solves twice (DISTANCE, then AZIMUTH); both come from the same solution. */
HH_HD void hh_geo_inverse(double lat1, double lon1, double lat2, double lon2, double *s12_out, double *azi1_out) {
    double lon12s;
    double lon12 = hh_ang_diff(lon1, lon2, &lon12s);
    double lonsign = hh_copysign(1.0, lon12);
    lon12 = lonsign * lon12;
    lon12s = lonsign * lon12s;
    double lam12 = lon12 * HH_DEG2RAD;
    double slam12, clam12;
    hh_sincosde(lon12, lon12s, &slam12, &clam12);
    lon12s = (180.0 - lon12) - lon12s;
    lat1 = hh_ang_round(lat1);
    lat2 = hh_ang_round(lat2);
    double swapp = hh_fabs(lat1) < hh_fabs(lat2) ? -1.0 : 1.0;
    if (swapp < 0.0) {
        lonsign = -lonsign;
        double t = lat1; lat1 = lat2; lat2 = t;
    }
    double latsign = hh_copysign(1.0, -lat1);
    lat1 *= latsign;
    lat2 *= latsign;
    double sbet1, cbet1, sbet2, cbet2;
    hh_sincosd(lat1, &sbet1, &cbet1);
    sbet1 *= HH_GEO_F1;
    hh_geo_norm(&sbet1, &cbet1);
    cbet1 = hh_max(HH_GEO_TINY, cbet1);
    hh_sincosd(lat2, &sbet2, &cbet2);
    sbet2 *= HH_GEO_F1;
    hh_geo_norm(&sbet2, &cbet2);
    cbet2 = hh_max(HH_GEO_TINY, cbet2);
    if (cbet1 < -sbet1) {
        if (cbet2 == cbet1) sbet2 = hh_copysign(sbet1, sbet2);
    } else {
        if (hh_fabs(sbet2) == -sbet1) cbet2 = cbet1;
    }
    double dn1 = hh_sqrt(1.0 + HH_GEO_EP2 * sbet1 * sbet1);
    double dn2 = hh_sqrt(1.0 + HH_GEO_EP2 * sbet2 * sbet2);
    int meridian = (lat1 == -90.0) || (slam12 == 0.0);
    double s12x = 0.0, salp1 = 0.0, calp1 = 1.0, salp2 = 0.0, calp2 = 1.0;
    if (meridian) {
        calp1 = clam12; salp1 = slam12; calp2 = 1.0; salp2 = 0.0;
        double ssig1 = sbet1, csig1 = calp1 * cbet1, ssig2 = sbet2, csig2 = calp2 * cbet2;
        double sig12 = hh_atan2(hh_max(0.0, csig1 * ssig2 - ssig1 * csig2) + 0.0, csig1 * csig2 + ssig1 * ssig2);
        double m12x;
        hh_geo_lengths(HH_GEO_N, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12x, &m12x);
        if (sig12 < 1.0 || m12x >= 0.0) {
            if (sig12 < 3.0 * HH_GEO_TINY || (sig12 < HH_GEO_TOL0 && (s12x < 0.0 || m12x < 0.0))) s12x = 0.0;
            s12x *= HH_GEO_B;
        } else {
            meridian = 0;
        }
    }
    if (!meridian && sbet1 == 0.0 && lon12s >= HH_GEO_F * 180.0) {
        calp1 = calp2 = 0.0;
        salp1 = salp2 = 1.0;
        s12x = HH_GEO_A * lam12;
    } else if (!meridian) {
        /* starting point (Karney Sec. 5), short-line branch included */
        double sig12 = -1.0, dnm = 1.0;
        double sbet12 = sbet2 * cbet1 - cbet2 * sbet1;
        double cbet12 = cbet2 * cbet1 + sbet2 * sbet1;
        double sbet12a = sbet2 * cbet1 + cbet2 * sbet1;
        int shortline = cbet12 >= 0.0 && sbet12 < 0.5 && cbet2 * lam12 < 0.5;
        double somg12, comg12;
        if (shortline) {
            double sb = sbet1 + sbet2, cb = cbet1 + cbet2;
            double sbetm2 = sb * sb;
            sbetm2 /= sbetm2 + cb * cb;
            dnm = hh_sqrt(1.0 + HH_GEO_EP2 * sbetm2);
            double omg12 = lam12 / (HH_GEO_F1 * dnm);
            hh_sincos(omg12, &somg12, &comg12);
        } else {
            somg12 = slam12;
            comg12 = clam12;
        }
        salp1 = cbet2 * somg12;
        calp1 = comg12 >= 0.0 ? sbet12 + cbet2 * sbet1 * somg12 * somg12 / (1.0 + comg12)
                              : sbet12a - cbet2 * sbet1 * somg12 * somg12 / (1.0 - comg12);
        double ssig12 = hh_hypot(salp1, calp1);
        double csig12 = sbet1 * sbet2 + cbet1 * cbet2 * comg12;
        if (shortline && ssig12 < HH_GEO_ETOL2) {
            salp2 = cbet1 * somg12;
            calp2 = sbet12 - cbet1 * sbet2 * (comg12 >= 0.0 ? somg12 * somg12 / (1.0 + comg12) : 1.0 - comg12);
            hh_geo_norm(&salp2, &calp2);
            sig12 = hh_atan2(ssig12, csig12);
        }
        if (!(salp1 <= 0.0)) {
            hh_geo_norm(&salp1, &calp1);
        } else {
            salp1 = 1.0;
            calp1 = 0.0;
        }
        if (sig12 >= 0.0) {
            s12x = sig12 * HH_GEO_B * dnm;
        } else {
            /* Newton on alp1 (Karney eq. 45-46) */
            hh_geo_l12 L;
            int numit = 0, tripn = 0;
            for (;;) {
                hh_geo_lambda12(sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam12, clam12, &L);
                double v = L.lam12;
                if (!(hh_fabs(v) >= (tripn ? 8.0 : 1.0) * HH_GEO_TOL0)) break;
                numit++;
                if (numit >= HH_GEO_MAXIT || !(L.dlam12 > 0.0)) break;
                double dalp1 = -v / L.dlam12;
                double sdalp1, cdalp1;
                hh_sincos(dalp1, &sdalp1, &cdalp1);
                double nsalp1 = salp1 * cdalp1 + calp1 * sdalp1;
                if (!(nsalp1 > 0.0 && hh_fabs(dalp1) < HH_PI)) break;
                calp1 = calp1 * cdalp1 - salp1 * sdalp1;
                salp1 = nsalp1;
                hh_geo_norm(&salp1, &calp1);
                tripn = hh_fabs(v) <= 16.0 * HH_GEO_TOL0;
            }
            double s12b, m12b;
            hh_geo_lengths(L.eps, L.sig12, L.ssig1, L.csig1, dn1, L.ssig2, L.csig2, dn2, &s12b, &m12b);
            s12x = s12b * HH_GEO_B;
            salp2 = L.salp2;
            calp2 = L.calp2;
        }
    }
    if (swapp < 0.0) {
        salp1 = salp2;
        calp1 = calp2;
    }
    salp1 *= swapp * lonsign;
    calp1 *= swapp * latsign;
    *s12_out = 0.0 + s12x;
    *azi1_out = hh_atan2d(salp1, calp1);
}

#endif /* HH_GEODESIC_H */
