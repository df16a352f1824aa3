/*
 * po_detmath.h -- TEST INFRASTRUCTURE.  The oracle's own plain-C statement of the
 * fdlibm (Sun Microsystems 1993) exp / sin / cos algorithms (e_exp.c, k_sin.c,
 * k_cos.c, e_rem_pio2.c medium path), used when po_scene.detMath != 0 so that a
 * cost value does not depend on which libm the host happens to have (the
 * reference calls the platform libm: patch.cpp:1034,1037,623; utility.h:26-28).
 * tests/test_detmath.py checks these against glibc (<= 1 ulp) and, bit for bit,
 * against the product's independent copy (pais_mvs_amd/csrc/pais_detmath.hpp).
 */
#ifndef PO_DETMATH_H
#define PO_DETMATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>


static inline uint64_t po_d2u(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
static inline double po_u2d(uint64_t u)
{
    double x;
    memcpy(&x, &u, 8);
    return x;
}
static inline int32_t po_hi_word(double x) { return (int32_t)(po_d2u(x) >> 32); }

static inline double po_det_exp(double x)
{
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02;
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double twom1000 = 9.33263618503218878990e-302, huge = 1.0e+300;
    double hi = 0, lo = 0, c, t, y;
    int32_t k = 0;
    uint32_t hx = (uint32_t)po_hi_word(x);
    const int xsb = (int)((hx >> 31) & 1);
    hx &= 0x7fffffff;
    if (hx >= 0x40862E42) {
        if (hx >= 0x7ff00000) {
            if (x != x) return x + x;             // NaN
            return (xsb == 0) ? x : 0.0;          // exp(+-inf)
        }
        if (x > o_threshold) return huge * huge;
        if (x < u_threshold) return twom1000 * twom1000;
    }
    if (hx > 0x3fd62e42) {
        if (hx < 0x3FF0A2B2) {
            hi = xsb ? (x + ln2HI) : (x - ln2HI);
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
            t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000) {
        return 1.0 + x;
    } else {
        k = 0;
    }
    t = x * x;
    c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) {
        return po_u2d(po_d2u(y) + ((uint64_t)(int64_t)k << 52));
    }
    return po_u2d(po_d2u(y) + ((uint64_t)(int64_t)(k + 1000) << 52)) * twom1000;
}

static inline double po_det_ksin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
static inline double po_det_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    int32_t ix = po_hi_word(x) & 0x7fffffff;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000)
        qx = 0.28125;
    else
        qx = po_u2d(((uint64_t)(uint32_t)(ix - 0x00200000)) << 32);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
// argument reduction, |x| < 2^19*pi/2 (angles here are O(10)); returns n mod 4 and y0+y1 = x - n*pi/2
static inline int po_det_rem_pio2(double x, double *y0, double *y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    int32_t hx = po_hi_word(x);
    int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3fe921fb) {
        *y0 = x;
        *y1 = 0;
        return 0;
    }
    double t = x < 0 ? -x : x;
    int32_t n = (int32_t)(t * invpio2 + 0.5);
    double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    int32_t j = ix >> 20;
    double a = r - w;
    int32_t i = j - ((po_hi_word(a) >> 20) & 0x7ff);
    if (i > 16) {
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - ((po_hi_word(a) >> 20) & 0x7ff);
        if (i > 49) {
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            a = r - w;
        }
    }
    double b = (r - a) - w;
    if (hx < 0) {
        *y0 = -a;
        *y1 = -b;
        return (-n) & 3;
    }
    *y0 = a;
    *y1 = b;
    return n & 3;
}
static inline double po_det_sin(double x)
{
    if (x != x || x - x != 0.0) return x - x; // NaN / inf
    double y0, y1;
    int n = po_det_rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return po_det_ksin(y0, y1, 1);
    case 1: return po_det_kcos(y0, y1);
    case 2: return -po_det_ksin(y0, y1, 1);
    default: return -po_det_kcos(y0, y1);
    }
}
static inline double po_det_cos(double x)
{
    if (x != x || x - x != 0.0) return x - x;
    double y0, y1;
    int n = po_det_rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return po_det_kcos(y0, y1);
    case 1: return -po_det_ksin(y0, y1, 1);
    case 2: return -po_det_kcos(y0, y1);
    default: return po_det_ksin(y0, y1, 1);
    }
}


/* exp of the cost weights in kernel-arithmetic mode: the same operations as pais::det_exp_poly
 * (pais_mvs_amd/csrc/pais_detmath.hpp) -- Cody-Waite reduction with fdlibm's split of ln2, Taylor polynomial of degree
 * 13 in Horner form with fma, ldexp.  <= 1 ulp from glibc. */
static inline double po_det_exp_poly(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const int ok = (x > -746.0) && (x < 710.0);
    const double xs = ok ? x : 0.0;
    const double k = rint(xs * invln2);
    double r = fma(-k, ln2HI, xs);
    r = fma(-k, ln2LO, r);
    double q = 1.6059043836821613e-10;
    q = fma(q, r, 2.08767569878681e-09);
    q = fma(q, r, 2.505210838544172e-08);
    q = fma(q, r, 2.755731922398589e-07);
    q = fma(q, r, 2.7557319223985893e-06);
    q = fma(q, r, 2.48015873015873e-05);
    q = fma(q, r, 1.984126984126984e-04);
    q = fma(q, r, 1.388888888888889e-03);
    q = fma(q, r, 8.333333333333333e-03);
    q = fma(q, r, 4.1666666666666664e-02);
    q = fma(q, r, 1.6666666666666666e-01);
    q = fma(q, r, 0.5);
    const double s = fma(r * r, q, r);
    const double y = ldexp(1.0 + s, (int)k);
    return ok ? y : ((x != x) ? (x + x) : ((x < 0.0) ? 0.0 : (x + x) * 1.0e300 * 1.0e300));
}

#endif
