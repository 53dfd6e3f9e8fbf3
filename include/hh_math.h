/*
 * hh_math.h — bit-reproducible FP64 elementary functions for the air-combat world.
 *
 * Why this exists: the parity bar is "integer hit/detect masks bit-exact, floats <= 1e-5"
 * between the CPU oracle and the gfx950 kernels.  Masks are float-vs-threshold compares, so
 * the only way to make them *guaranteed* identical (not just statistically identical) is for
 * host and device to execute the same IEEE-754 operation sequence.  glibc libm and the
 * device's ocml differ in the last ulp, so neither is used: everything here is built from
 * +,-,*,/,sqrt,fma (all correctly rounded on x86-64 and on gfx950) with explicit fma() calls,
 * and both sides are compiled with -ffp-contract=off.
 *
 * It is also the cheap path on CDNA4: arguments on this workload are bounded (|angle| < 1e3
 * degrees, |radians| < 1e2), so range reduction is a two-term Cody-Waite step instead of
 * ocml's generic Payne-Hanek machinery.
 *
 * Polynomial kernels are the classic minimax fits used by every fdlibm-lineage libm
 * (accuracy < 1 ulp on the reduced range); tests/test_math.py pins each function against
 * mpmath at <= 2 ulp over the ranges this workload uses.
 *
 * C99-compatible so that the plain-C oracle (gcc) and the HIP kernels (hipcc) share it.
 */
#ifndef HH_MATH_H
#define HH_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define HH_HD __host__ __device__ __forceinline__
#else
#define HH_HD static inline
#endif

#define HH_PI 3.14159265358979323846
#define HH_DEG2RAD (HH_PI / 180.0)   /* CPython: degToRad = pi / 180.0 */
#define HH_RAD2DEG (180.0 / HH_PI)   /* CPython: radToDeg = 180.0 / pi */

/* ---- primitives that map to single correctly-rounded instructions on both targets ---- */
HH_HD double hh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* sqrt: the host's instruction is correctly rounded.  On gfx950 there is no such instruction and the compiler expands the operator
 * into v_rsq_f64 + a Goldschmidt / Newton sequence that IS correctly rounded, wrapped in a rescaling for operands below 2^-767 (two
 * compares, two ldexp, two selects) that no operand of the step can need: squares of sines and cosines, squared separations in
 * degrees (zero, or >= 1e-32) and 1 - e^2 sin^2.  The device form is that same sequence without the rescaling — identical bits for
 * x = 0, x = +inf and x >= 2^-767, seven instructions fewer per root (tests/test_gpu_math.py compares it with the host on the GPU). */
#if defined(__HIP_DEVICE_COMPILE__)
HH_HD double hh_sqrt(double x) /* x = 0 or x >= 2^-767 */ {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d1, h, g);
    return ((x == 0.0) | (x == __builtin_inf())) ? x : g;
}
#else
HH_HD double hh_sqrt(double x) { return __builtin_sqrt(x); }
#endif
HH_HD double hh_fabs(double x) { return __builtin_fabs(x); }
HH_HD double hh_floor(double x) { return __builtin_floor(x); }
HH_HD double hh_trunc(double x) { return __builtin_trunc(x); }
HH_HD double hh_rint(double x) { return __builtin_rint(x); } /* ties-to-even, like Python 3 round() */
HH_HD double hh_copysign(double x, double y) { return __builtin_copysign(x, y); }
HH_HD double hh_min(double a, double b) { return a < b ? a : b; }
HH_HD double hh_max(double a, double b) { return a > b ? a : b; }
/* clip(x, lo, hi), lo <= hi, as max then min: one v_max_f64 + one v_min_f64 on the GPU (a single clamped instruction for [0, 1])
 * instead of two compares and four selects.  The host form picks the same operand in every case the device instructions do —
 * NaN -> lo, -0.0 against a +0.0 bound -> the bound — so both sides still produce the same bits; against numpy.clip the only
 * difference is the sign of a zero result (and NaN, which the step never produces). */
#if defined(__HIP_DEVICE_COMPILE__)
HH_HD double hh_clip(double x, double lo, double hi) { return __builtin_fmin(__builtin_fmax(x, lo), hi); }
#else
HH_HD double hh_clip(double x, double lo, double hi) {
    const double y = x > lo ? x : lo;
    return y < hi ? y : hi;
}
#endif
/* hypot without scaling: operands here are O(1e-4..1e2), no overflow/underflow risk */
HH_HD double hh_hypot(double x, double y) { return hh_sqrt(x * x + y * y); }

/* ---- exact fmod / Python float modulo / IEEE remainder for small quotients ----
 * |x/m| < 2^30 on this path (headings, longitudes).  The true fmod result is always exactly
 * representable, so one fma(-q, m, x) with a +-1 correction of q is exact. */
/* The trial quotient only has to be within one of floor(|x| / |m|): the exact fma remainder and the two fix-ups
 * absorb an off-by-one, so it is taken by reciprocal multiplication (the reciprocal folds to a constant for the
 * literal moduli 360 / 359 this code uses; |x| / |m| < 2^50). */
HH_HD double hh_fmod(double x, double m) {
    double ax = hh_fabs(x), am = hh_fabs(m);
    /* no early return for |x| < |m|: the general path gives x back there (q = 0, or q = 1 undone by the first fix-up; the sign
     * of a zero survives the copysign), and on the GPU a skipped branch costs more than these five instructions */
    double q = hh_floor(ax * (1.0 / am));
    double r = hh_fma(-q, am, ax);
    if (r < 0.0) r += am;
    if (r >= am) r -= am;
    return hh_copysign(r, x);
}

/* x / c for a divisor c known ahead (a literal, or a configuration value whose reciprocal rc = 1.0 / c was taken
 * once): Markstein's correction step — q0 = x * rc is within an ulp or two of the quotient, the fma residual
 * r = x - q0 * c is exact, and q0 + r * rc rounds to the correctly rounded x / c (tests/test_math.py compares it
 * with the division operator on 10^7 operands per divisor).  3 dependent operations instead of the ~12 of an
 * IEEE division; operands here are normal, moderate numbers (degrees, knots, counts). */
HH_HD double hh_div_known(double x, double c, double rc) {
    double q = x * rc;
    double r = hh_fma(-q, c, x);
    return hh_fma(r, rc, q);
}
#define HH_DIVC(x, c) hh_div_known((x), (c), 1.0 / (c))

/* CPython float_rem (Objects/floatobject.c semantics): result has the sign of m */
HH_HD double hh_pymod(double x, double m) {
    double r = hh_fmod(x, m);
    if (r != 0.0) {
        if ((m < 0.0) != (r < 0.0)) r += m;
    } else {
        r = hh_copysign(0.0, m);
    }
    return r;
}

/* x % 360.0 / x % 359.0 where the operand is a heading plus or minus less than a turn — every call of the step.  Inside
 * [-m, 2m) the value of hh_pymod is one exact subtraction (x - m for m <= x < 2m: Sterbenz), the same rounded addition CPython
 * makes (x + m for x < 0), or x itself (+ 0.0 turns the -0.0 of an empty remainder into CPython's +0.0): ~10 instructions
 * instead of ~37, no branch.  PRECONDITION -m <= x < 2m: headings are kept in [0, 360) by the step itself (every write goes through
 * a modulo or the 0 <= h < 360 guard of _take_base_action, hh_set_state refuses others) and the operands here are a heading plus
 * or minus less than a turn; tests/test_math.py compares the two forms over the interval, its end points included, and the oracle
 * keeps calling the general form, so every parity run cross-checks the kernels' use of this one. */
HH_HD double hh_pymod_turn(double x, double m) /* m > 0, -m <= x < 2 m */ {
    const double lo = x + (x < 0.0 ? m : 0.0);
    return x >= m ? x - m : lo;
}
HH_HD double hh_pymod360(double x) { return hh_pymod_turn(x, 360.0); }
HH_HD double hh_pymod359(double x) { return hh_pymod_turn(x, 359.0); }

/* IEEE remainder(x, m): x - n*m with n = nearest integer to x/m, ties to even */
HH_HD double hh_remainder(double x, double m) {
    double am = hh_fabs(m);
    double r = hh_fmod(x, am);           /* |r| < am, sign of x */
    double ar = hh_fabs(r);
    double half = 0.5 * am;
    if (ar > half) {
        ar -= am;
    } else if (ar == half) {
        /* tie: choose even n */
        double q = hh_floor(hh_fabs(x) / am);
        double q2 = q * 0.5;
        if (q2 != hh_floor(q2)) ar -= am; /* q odd -> round up to even n = q+1 */
    }
    if (ar == 0.0) return hh_copysign(0.0, x);
    return x < 0.0 ? -ar : ar;
}

/* ---- sin/cos kernels on |x| <= pi/4 (+ tail y) ---- */
HH_HD double hh_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double r = hh_fma(z, hh_fma(z, hh_fma(z, hh_fma(z, S6, S5), S4), S3), S2);
    double v = z * x;
    /* x - ((z*(y/2 - v*r) - y) - v*S1), every product-sum as one fused operation */
    double t = hh_fma(-v, r, 0.5 * y);
    double u = hh_fma(z, t, -y);
    return x - hh_fma(-v, S1, u);
}

HH_HD double hh_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * hh_fma(z, hh_fma(z, hh_fma(z, hh_fma(z, hh_fma(z, C6, C5), C4), C3), C2), C1);
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + hh_fma(z, r, -(x * y)));
}

/* sin and cos of x radians, |x| < ~1e5 (two-term Cody-Waite reduction by pi/2) */
HH_HD void hh_sincos(double x, double *s, double *c) {
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    double fn = hh_rint(x * INVPIO2);
    double r = hh_fma(-fn, PIO2_1, x);
    double w = fn * PIO2_1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double ks = hh_ksin(y0, y1);
    double kc = hh_kcos(y0, y1);
    int n = (int)fn & 3;
    double ss = (n & 1) ? kc : ks;
    double cc = (n & 1) ? ks : kc;
    if (n == 1 || n == 2) cc = -cc;
    if (n == 2 || n == 3) ss = -ss;
    *s = ss;
    *c = cc;
}

/* ---- atan / atan2 ---- */
HH_HD double hh_atan_pos(double x) /* x >= 0 */ {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    /* argument reduction t = num/den on five intervals.  Every candidate is computed and chosen by selects on plain values, and the
     * special cases are selected at the end: on the GPU a region skipped under the exec mask costs ~14 instruction slots, more than
     * any of these arms; the chosen expressions, and so the bits, are those of the branching form */
    const double n0 = 2.0 * x - 1.0, d0 = 2.0 + x, n1 = x - 1.0, d1 = x + 1.0, n2 = x - 1.5, d2 = 1.0 + 1.5 * x;
    const int low = x < 0.4375, i0 = x < 0.6875, i1 = x < 1.1875, i2 = x < 2.4375;
    double num = -1.0, den = x, hi = 1.57079632679489655800e+00, lo = 6.12323399573676603587e-17;
    num = i2 ? n2 : num; den = i2 ? d2 : den; hi = i2 ? 9.82793723247329054082e-01 : hi; lo = i2 ? 1.39033110312309984516e-17 : lo;
    num = i1 ? n1 : num; den = i1 ? d1 : den; hi = i1 ? 7.85398163397448278999e-01 : hi; lo = i1 ? 3.06161699786838301793e-17 : lo;
    num = i0 ? n0 : num; den = i0 ? d0 : den; hi = i0 ? 4.63647609000806093515e-01 : hi; lo = i0 ? 2.26987774529616870924e-17 : lo;
    num = low ? x : num; den = low ? 1.0 : den; /* x / 1 is exact: t = x */
    double t = num / den;
    double z = t * t;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double r_low = t - t * (s1 + s2);
    double r_hi = hi - ((t * (s1 + s2) - lo) - t);
    const double r = low ? r_low : r_hi;
    const double r_huge = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return x > 1e300 ? r_huge : r;
}

HH_HD double hh_atan2(double y, double x) {
    const double PI_LO = 1.2246467991473531772e-16;
    /* main path first, special cases selected over it in reverse order of the branching form's tests (same values for every
     * operand pair: the main path's garbage for zeros / NaNs is never the one chosen) */
    const double z = hh_atan_pos(hh_fabs(y / x));
    const double zn = -z, zz = HH_PI - (z - PI_LO), zzn = -zz;
    const int ypos = y > 0.0;
    const double r_right = ypos ? z : zn, r_left = ypos ? zz : zzn;
    double r = x > 0.0 ? r_right : r_left;
    const double r_x0 = hh_copysign(0.5 * HH_PI, y);
    r = x == 0.0 ? r_x0 : r;
    const int x_plus = (x > 0.0) | ((x == 0.0) & (hh_copysign(1.0, x) > 0.0));
    const double r_y0b = hh_copysign(HH_PI, y);
    const double r_y0 = x_plus ? y : r_y0b; /* +-0 */
    r = y == 0.0 ? r_y0 : r;
    const double r_nan = x + y;
    return ((x != x) | (y != y)) ? r_nan : r;
}

/* ---- acos ---- */
/* acos(a) = sqrt(1 - a) * P(2a - 1) on [0, 1] (acos(x)/sqrt(1-x) is analytic there; degree-19 interpolant, truncation
 * 3e-17 relative, coefficients from tools/gen_acos_coeffs.py), acos(-a) = pi - acos(a).  One square root and 21 fused
 * multiply-adds in two independent Horner chains (even / odd powers): no division, straight-line, and 1 - a is exact for
 * a >= 1/2, so small angles keep their relative accuracy.  Within 2 ulp of the correctly rounded value (tests/test_math.py). */
HH_HD double hh_acos(double x) /* |x| <= 1 */ {
    const double P0 = 1.48096097938612203e+00, P1 = -7.60160912346649897e-02, P2 = 1.10293133179791107e-02,
                 P3 = -2.14913585901438647e-03, P4 = 4.82054100561475694e-04, P5 = -1.17412504152359799e-04,
                 P6 = 3.01871701458175376e-05, P7 = -8.06353963468094371e-06, P8 = 2.21601896416617581e-06,
                 P9 = -6.22532301137391218e-07, P10 = 1.77975683093472170e-07, P11 = -5.16095474235324628e-08,
                 P12 = 1.51278938015423088e-08, P13 = -4.48328265569854430e-09, P14 = 1.36250811190112031e-09,
                 P15 = -4.10493165700480771e-10, P16 = 1.05376345601740400e-10, P17 = -3.20146929819632065e-11,
                 P18 = 1.87618285989296095e-11, P19 = -5.79529090405995285e-12;
    const double PI_LO = 1.22464679914735317723e-16; /* pi - HH_PI */
    double a = hh_fabs(x);
    a = a > 1.0 ? 1.0 : a; /* callers clip; keeps the root real */
    const double t = hh_fma(2.0, a, -1.0);
    const double t2 = t * t;
    double pe = hh_fma(t2, P18, P16), po = hh_fma(t2, P19, P17);
    pe = hh_fma(t2, pe, P14); po = hh_fma(t2, po, P15);
    pe = hh_fma(t2, pe, P12); po = hh_fma(t2, po, P13);
    pe = hh_fma(t2, pe, P10); po = hh_fma(t2, po, P11);
    pe = hh_fma(t2, pe, P8); po = hh_fma(t2, po, P9);
    pe = hh_fma(t2, pe, P6); po = hh_fma(t2, po, P7);
    pe = hh_fma(t2, pe, P4); po = hh_fma(t2, po, P5);
    pe = hh_fma(t2, pe, P2); po = hh_fma(t2, po, P3);
    pe = hh_fma(t2, pe, P0); po = hh_fma(t2, po, P1);
    const double r = hh_sqrt(1.0 - a) * hh_fma(t, po, pe);
    return x < 0.0 ? (HH_PI - r) + PI_LO : r;
}

/* ---- degree helpers used by the geodesic layer (Karney 2013, Sec. 6 implementation notes:
 * exact quadrant reduction so that cardinal headings give exact zeros) ---- */

/* AngRound: make tiny angles (|x| < 1/16 deg) round on a fixed grid so -0/underflow behave */
HH_HD double hh_ang_round(double x) {
    const double z = 1.0 / 16.0;
    double y = hh_fabs(x);
    double t = z - y; /* two separately rounded steps (no fast-math: not simplified) */
    y = y < z ? z - t : y;
    return hh_copysign(y, x);
}

/* sin, cos of x degrees with exact reduction to [-45, 45] */
HH_HD void hh_sincosd(double x, double *sinx, double *cosx) {
    double r = hh_fmod(x, 360.0);
    double qf = hh_rint(r / 90.0);
    int q = (int)qf;
    r -= 90.0 * qf;
    r *= HH_DEG2RAD;
    double s = hh_ksin(r, 0.0), c = hh_kcos(r, 0.0);
    double ss, cc;
    switch ((unsigned)q & 3u) {
        case 0u: ss = s; cc = c; break;
        case 1u: ss = c; cc = -s; break;
        case 2u: ss = -s; cc = -c; break;
        default: ss = -c; cc = s; break;
    }
    cc = 0.0 + cc;
    if (ss == 0.0) ss = hh_copysign(ss, x);
    *sinx = ss;
    *cosx = cc;
}

/* sin, cos of x degrees for |x| <= 360 without divisions (used by the short-step move, where the result
 * only has to be accurate, not identical to the Karney helper above): q = nearest multiple of 90 */
HH_HD void hh_sincosd_small(double x, double *sinx, double *cosx) {
    double qf = hh_rint(x * (1.0 / 90.0));
    int q = (int)qf;
    double r = hh_fma(-90.0, qf, x) * HH_DEG2RAD; /* exact remainder in [-45.0000001, 45.0000001] */
    double s = hh_ksin(r, 0.0), c = hh_kcos(r, 0.0);
    double ss = (q & 1) ? c : s;
    double cc = (q & 1) ? s : c;
    if (((q + 1) & 2) != 0) cc = -cc; /* q mod 4 in {1, 2} */
    if ((q & 2) != 0) ss = -ss;       /* q mod 4 in {2, 3} */
    *sinx = ss;
    *cosx = cc;
}

/* sin, cos of (x + t) degrees where t is a small correction (AngDiff's error term) */
HH_HD void hh_sincosde(double x, double t, double *sinx, double *cosx) {
    double qf = hh_rint(x / 90.0);
    int q = (int)qf;
    double r = x - 90.0 * qf;
    r = hh_ang_round(r + t) * HH_DEG2RAD;
    double s = hh_ksin(r, 0.0), c = hh_kcos(r, 0.0);
    double ss, cc;
    switch ((unsigned)q & 3u) {
        case 0u: ss = s; cc = c; break;
        case 1u: ss = c; cc = -s; break;
        case 2u: ss = -s; cc = -c; break;
        default: ss = -c; cc = s; break;
    }
    cc = 0.0 + cc;
    if (ss == 0.0) ss = hh_copysign(ss, x);
    *sinx = ss;
    *cosx = cc;
}

/* atan2 in degrees with octant reduction (result in [-180, 180]) */
HH_HD double hh_atan2d(double y, double x) {
    int q = 0;
    if (hh_fabs(y) > hh_fabs(x)) { double t = x; x = y; y = t; q = 2; }
    if (x < 0.0) { x = -x; q += 1; }
    double ang = hh_atan2(y, x) * HH_RAD2DEG;
    switch (q) {
        case 1: ang = hh_copysign(180.0, y) - ang; break;
        case 2: ang = 90.0 - ang; break;
        case 3: ang = -90.0 + ang; break;
        default: break;
    }
    return ang;
}

/* error-free sum: s = fl(u+v), *t = exact (u+v) - s */
HH_HD double hh_two_sum(double u, double v, double *t) {
    double s = u + v;
    double up = s - v;
    double vpp = s - up;
    up -= u;
    vpp -= v;
    *t = s != 0.0 ? 0.0 - (up + vpp) : s;
    return s;
}

/* AngNormalize: reduce to [-180, 180] */
HH_HD double hh_ang_normalize(double x) {
    double y = hh_remainder(x, 360.0);
    return hh_fabs(y) == 180.0 ? hh_copysign(180.0, x) : y;
}

/* AngDiff: d = y - x reduced to [-180,180], *e = rounding error of d */
HH_HD double hh_ang_diff(double x, double y, double *e) {
    double t;
    double d = hh_two_sum(hh_remainder(-x, 360.0), hh_remainder(y, 360.0), &t);
    d = hh_two_sum(hh_remainder(d, 360.0), t, &t);
    if (d == 0.0 || hh_fabs(d) == 180.0) d = hh_copysign(d, t == 0.0 ? y - x : -t);
    *e = t;
    return d;
}

/* Python's round(x, 3) (used by the scripted opponent's turn-direction test) */
HH_HD double hh_round3(double x) {
    double y = x * 1000.0;
    double z = hh_rint(y);
    return HH_DIVC(z, 1000.0); /* = z / 1000.0 for every integer the callers reach (sines and cosines: |z| <= 1000; tests/test_math.py
                                * checks |z| <= 10^6): three multiply-adds instead of an IEEE division */
}

#endif /* HH_MATH_H */
