/*
 * po_pmath.h — portable double-precision sin / cos / atan / atan2 (and x^1.5) built from IEEE +, -, *, / and sqrt only.
 *
 * Why: the map stages either side of the QP (corridor bounds, DP lattice search, re-sampling, projections: SURVEY.md §8f) take thresholds and ties on
 * quantities that pass through sin / cos / atan2.  A GPU's libm and glibc's both stay below one ulp of error but do not round identically, so device and
 * CPU checker could differ in the last bit (and, once in a long while, in an index).  These routines are the SAME sequence of IEEE operations wherever
 * they are compiled (no FMA contraction: the translation units that use them are built with -ffp-contract=off), so the HIP kernels and the oracle's
 * "portable math" mode (po_oracle_set_portable_math) agree bit for bit.  They are not meant to beat libm: accuracy < 1 ulp (tests/test_pmath.py measures
 * it against glibc), within one ulp of glibc's result, which is what the oracle keeps using when it is pinned against the reference's own binaries.
 *
 * Method (the classic one, cf. Sun's fdlibm): Cody-Waite reduction by pi/2 in three pieces, minimax polynomials for sin and cos on [-pi/4, pi/4],
 * atan by argument reduction to four intervals + an odd polynomial, atan2 by quadrant.  Valid for finite arguments, |x| < 2^19 * pi/2 for sin / cos
 * (angles here are headings of a few radians).
 */
#ifndef PO_PMATH_H_
#define PO_PMATH_H_

#if defined(__HIPCC__)
#define PO_PM_FN __host__ __device__ static inline
#else
#define PO_PM_FN static inline
#endif
#include <string.h>

PO_PM_FN int po_pm_hi(double x) {  /* high 32 bits of the IEEE pattern */
    long long b;
    memcpy(&b, &x, sizeof b);
    return (int)(b >> 32);
}
PO_PM_FN double po_pm_from_hi(int hi) {
    long long b = ((long long)hi) << 32;
    double x;
    memcpy(&x, &b, sizeof x);
    return x;
}
PO_PM_FN double po_pm_fabs(double x) { return x < 0 ? -x : x; }

/* sin on [-pi/4, pi/4], x + y the reduced argument (y the tail), iy = 0 when y is exactly 0 */
PO_PM_FN double po_pm_ksin(double x, double y, int iy) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const int ix = po_pm_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) { if ((int)x == 0) return x; }  /* |x| < 2^-27 */
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
PO_PM_FN double po_pm_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int ix = po_pm_hi(x) & 0x7fffffff;
    if (ix < 0x3e400000) { if ((int)x == 0) return 1.0; }
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000) qx = 0.28125;
    else qx = po_pm_from_hi(ix - 0x00200000);  /* about x / 4 */
    const double hz = 0.5 * z - qx, a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
/* x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4; returns n (sign included).  |x| < 2^19 * pi/2. */
PO_PM_FN int po_pm_rem_pio2(double x, double *y0, double *y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11;
    const double p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21;
    const double p3 = 2.02226624871116645580e-21, p3t = 8.47842766036889956997e-32;
    const int hx = po_pm_hi(x), ix = hx & 0x7fffffff;
    const double t = po_pm_fabs(x);
    const int n = (int)(t * invpio2 + 0.5);
    const double fn = (double)n;
    double r = t - fn * p1, w = fn * p1t;
    double a0 = r - w;
    const int j = ix >> 20;
    int i = j - ((po_pm_hi(a0) >> 20) & 0x7ff);
    if (i > 16) {  /* a second piece of pi/2 is needed */
        double tt = r;
        w = fn * p2;
        r = tt - w;
        w = fn * p2t - ((tt - r) - w);
        a0 = r - w;
        i = j - ((po_pm_hi(a0) >> 20) & 0x7ff);
        if (i > 49) {  /* and a third */
            tt = r;
            w = fn * p3;
            r = tt - w;
            w = fn * p3t - ((tt - r) - w);
            a0 = r - w;
        }
    }
    const double a1 = (r - a0) - w;
    if (hx < 0) { *y0 = -a0; *y1 = -a1; return -n; }
    *y0 = a0; *y1 = a1;
    return n;
}
PO_PM_FN double po_psin(double x) {
    const int ix = po_pm_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return po_pm_ksin(x, 0.0, 0);
    double y0, y1;
    const int n = po_pm_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return po_pm_ksin(y0, y1, 1);
        case 1: return po_pm_kcos(y0, y1);
        case 2: return -po_pm_ksin(y0, y1, 1);
        default: return -po_pm_kcos(y0, y1);
    }
}
PO_PM_FN double po_pcos(double x) {
    const int ix = po_pm_hi(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return po_pm_kcos(x, 0.0);
    double y0, y1;
    const int n = po_pm_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return po_pm_kcos(y0, y1);
        case 1: return -po_pm_ksin(y0, y1, 1);
        case 2: return -po_pm_kcos(y0, y1);
        default: return po_pm_ksin(y0, y1, 1);
    }
}
PO_PM_FN double po_patan(double x) {
    const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01, hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
    const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17, lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
    const double T0 = 3.33333333333329318027e-01, T1 = -1.99999999998764832476e-01, T2 = 1.42857142725034663711e-01, T3 = -1.11111104054623557880e-01,
                 T4 = 9.09088713343650656196e-02, T5 = -7.69187620504482999495e-02, T6 = 6.66107313738753120669e-02, T7 = -5.83357013379057348645e-02,
                 T8 = 4.97687799461593236017e-02, T9 = -3.65315727442169155270e-02, T10 = 1.62858201153657823623e-02;
    const int hx = po_pm_hi(x), ix = hx & 0x7fffffff;
    if (ix >= 0x44100000) return hx > 0 ? hi3 + lo3 : -hi3 - lo3;  /* |x| >= 2^66 */
    if (ix < 0x3e200000) return x;
    /* argument reduction: the interval picks numerator and denominator, ONE division follows (below 0.4375 it is x / 1, exact).  Same operations per
       interval as the textbook's branch-per-interval form; on a GPU the lanes of a wave fall into different intervals, and four branches with a
       division each ran one after the other. */
    const double ax = po_pm_fabs(x);
    const int id = ix < 0x3fdc0000 ? -1 : (ix < 0x3fe60000 ? 0 : (ix < 0x3ff30000 ? 1 : (ix < 0x40038000 ? 2 : 3)));
    const double num = id < 0 ? x : (id == 0 ? 2.0 * ax - 1.0 : (id == 1 ? ax - 1.0 : (id == 2 ? ax - 1.5 : -1.0)));
    const double den = id < 0 ? 1.0 : (id == 0 ? 2.0 + ax : (id == 1 ? ax + 1.0 : (id == 2 ? 1.0 + 1.5 * ax : ax)));
    x = num / den;
    const double z = x * x, w = z * z;
    const double s1 = z * (T0 + w * (T2 + w * (T4 + w * (T6 + w * (T8 + w * T10)))));
    const double s2 = w * (T1 + w * (T3 + w * (T5 + w * (T7 + w * T9))));
    if (id < 0) return x - x * (s1 + s2);
    const double ahi = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3));
    const double alo = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
    const double r = ahi - ((x * (s1 + s2) - alo) - x);
    return hx < 0 ? -r : r;
}
PO_PM_FN double po_patan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_o_2 = 1.5707963267948965580E+00, pi_lo = 1.2246467991473531772E-16;
    const int hx = po_pm_hi(x), hy = po_pm_hi(y);
    const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (x == 1.0) return po_patan(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  /* 2 * sign(x) + sign(y) */
    if (y == 0.0) return m == 0 || m == 1 ? y : (m == 2 ? pi : -pi);
    if (x == 0.0) return hy < 0 ? -pi_o_2 : pi_o_2;
    const int k = (iy - ix) >> 20;
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;       /* |y / x| > 2^60 */
    else if (hx < 0 && k < -60) z = 0.0;        /* |y| / x < -2^60 */
    else z = po_patan(po_pm_fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}
/* x^1.5 for x >= 0 (curvature denominators): x * sqrt(x), two roundings, within one ulp of pow(x, 1.5) */
#if defined(__HIPCC__)
#define PO_PM_SQRT(x) sqrt(x)
#else
#include <math.h>
#define PO_PM_SQRT(x) sqrt(x)
#endif
PO_PM_FN double po_ppow15(double x) { return x * PO_PM_SQRT(x); }

#endif /* PO_PMATH_H_ */
