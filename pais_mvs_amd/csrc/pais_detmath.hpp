// pais_detmath.hpp -- reproducible exp / sin / cos for the PAIS hot path.
//
// The PSO that drives patch refinement is chaotic at the last bit: converged
// particles sit within an ulp of their personal best, so `fitness < pBestFitness`
// (psosolver.cpp:128) is routinely decided by rounding noise.
// To make the GPU path reproducible and checkable bit-for-bit against a CPU
// restatement, the elementary functions the cost calls (the reference calls the
// platform libm: exp at patch.cpp:1034,1037,623; sin/cos at utility.h:26-28) are
// evaluated with the classic fdlibm algorithms (Sun Microsystems, 1993:
// e_exp.c, k_sin.c, k_cos.c, e_rem_pio2.c medium path) written in plain IEEE-754
// double +,-,*,/ only -- identical bits on x86-64 and gfx950 as long as FP
// contraction is off.  Accuracy < 1 ulp, i.e. within libm-to-libm variability.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef PAIS_HD
#if defined(__HIPCC__)
#define PAIS_HD __host__ __device__ inline
#else
#define PAIS_HD inline
#endif
#endif

namespace pais {

PAIS_HD uint64_t d2u(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(x);
#else
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
#endif
}
PAIS_HD double u2d(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double x;
    memcpy(&x, &u, 8);
    return x;
#endif
}
PAIS_HD int32_t hi_word(double x) { return (int32_t)(d2u(x) >> 32); }

PAIS_HD double det_exp(double x)
{
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02;
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double twom1000 = 9.33263618503218878990e-302, huge = 1.0e+300;
    double hi = 0, lo = 0, c, t, y;
    int32_t k = 0;
    uint32_t hx = (uint32_t)hi_word(x);
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
        return u2d(d2u(y) + ((uint64_t)(int64_t)k << 52));
    }
    return u2d(d2u(y) + ((uint64_t)(int64_t)(k + 1000) << 52)) * twom1000;
}

// det_exp without lane-divergent branches, bit-identical to det_exp for every input (tests/test_detmath_and_devmath.py).
// fdlibm special-cases k = 0 (|x| <= 0.5 ln2) and k = +-1 (|x| < 1.5 ln2); both are instances of the general
// formula: with t = (double)k, hi = x - t*ln2HI and lo = t*ln2LO reproduce their hi / lo exactly (t = 0, +-1 multiply
// exactly), and for k = 0 the result 1 - ((x*c)/(c-2) - x) equals 1 - ((lo - (x*c)/(2-c)) - hi) because negating a
// divisor negates the correctly rounded quotient.  Only k itself needs fdlibm's threshold on the high word (the band
// just above 0.5 ln2 where the rounding formula would already give +-1) -- a select, not a branch.  |x| >= 700
// (denormal scaling, overflow, inf, NaN) takes the reference routine; the cost never gets there.
PAIS_HD double det_exp_bf(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const uint32_t hx0 = (uint32_t)hi_word(x);
    const uint32_t hx = hx0 & 0x7fffffff;
    if (hx >= 0x4085E000) return det_exp(x); // |x| >= 700, inf, NaN
    const double half = (hx0 >> 31) ? -0.5 : 0.5;
    const int32_t kr = (int32_t)(invln2 * x + half);
    const int32_t k = (hx > 0x3fd62e42) ? kr : 0;
    const double t = (double)k;
    const double hi = x - t * ln2HI;
    const double lo = t * ln2LO;
    const double xr = hi - lo;
    const double tt = xr * xr;
    const double c = xr - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    const double y = 1.0 - ((lo - (xr * c) / (2.0 - c)) - hi);
    const double scaled = u2d(d2u(y) + ((uint64_t)(int64_t)k << 52));
    return (hx < 0x3e300000) ? (1.0 + x) : scaled;
}

// exp for the cost weights (patch.cpp:1034,1037): no division, no lane-dependent branch, ~22 instructions instead of
// ~45.  Cody-Waite reduction x = k ln2 + r (|r| <= 0.3466, fdlibm's split of ln2), Taylor polynomial of degree 13 in
// Horner form with fma (truncation 4e-18), exact scaling with ldexp.  <= 1 ulp from glibc (tests); the oracle's
// kernel-arithmetic mode evaluates the very same operations (po_det_exp_poly).
PAIS_HD double det_exp_poly(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const bool ok = (x > -746.0) && (x < 710.0); // false for NaN too
    const double xs = ok ? x : 0.0;
    const double k = rint(xs * invln2);
    double r = fma(-k, ln2HI, xs);
    r = fma(-k, ln2LO, r);
    // Horner step q = q * r + c.  On the GPU the three-operand VOP3 form is forced: left alone the compiler emits
    // v_mov_b64 + v_fmac_f64 per step (the coefficients sit in VGPRs), i.e. 11 extra instructions per window pixel.
    // Round 5: the coefficient is the instruction's ONE scalar operand (an SGPR pair, set by two s_mov on the scalar unit,
    // which issues beside the VALU).  As VGPR operands the twelve coefficients were 24 registers that the compiler hoisted
    // out of the pixel loops and, in the register-bound kernels, spilled: the tile kernel re-loaded nine of them from
    // scratch per pixel, one `s_waitcnt vmcnt(0)` each -- a chain of memory round trips in the walk (profiles/r05_*).
#if defined(__HIP_DEVICE_COMPILE__)
#define PAIS_HORNER(c)                                                                        \
    {                                                                                         \
        const double c_ = (c);                                                                \
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(q), "v"(r), "s"(c_));                  \
    }
#else
#define PAIS_HORNER(c) q = fma(q, r, (c));
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    double q;                                // 1/13!  (moved in from the scalar unit at the point of use: as a loop-invariant VGPR
    {                                        //  constant it was hoisted out of the pixel loops and spilled with the others)
        const double c13 = 1.6059043836821613e-10;
        asm volatile("v_mov_b64 %0, %1" : "=v"(q) : "s"(c13));
    }
#else
    double q = 1.6059043836821613e-10;       // 1/13!
#endif
    PAIS_HORNER(2.08767569878681e-09)        // 1/12!
    PAIS_HORNER(2.505210838544172e-08)       // 1/11!
    PAIS_HORNER(2.755731922398589e-07)       // 1/10!
    PAIS_HORNER(2.7557319223985893e-06)      // 1/9!
    PAIS_HORNER(2.48015873015873e-05)        // 1/8!
    PAIS_HORNER(1.984126984126984e-04)       // 1/7!
    PAIS_HORNER(1.388888888888889e-03)       // 1/6!
    PAIS_HORNER(8.333333333333333e-03)       // 1/5!
    PAIS_HORNER(4.1666666666666664e-02)      // 1/4!
    PAIS_HORNER(1.6666666666666666e-01)      // 1/3!
    PAIS_HORNER(0.5)                         // 1/2!
#undef PAIS_HORNER
    const double s = fma(r * r, q, r);       // exp(r) - 1
    const double y = ldexp(1.0 + s, (int)k);
    // outside (-746, 710): 0 / +inf / NaN as exp gives them
    return ok ? y : ((x != x) ? (x + x) : ((x < 0.0) ? 0.0 : (x + x) * 1.0e300 * 1.0e300));
}

PAIS_HD double det_ksin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
PAIS_HD double det_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    int32_t ix = hi_word(x) & 0x7fffffff;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000)
        qx = 0.28125;
    else
        qx = u2d(((uint64_t)(uint32_t)(ix - 0x00200000)) << 32);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
// argument reduction, |x| < 2^19*pi/2 (angles here are O(10)); returns n mod 4 and y0+y1 = x - n*pi/2
PAIS_HD int det_rem_pio2(double x, double *y0, double *y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    int32_t hx = hi_word(x);
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
    int32_t i = j - ((hi_word(a) >> 20) & 0x7ff);
    if (i > 16) {
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - ((hi_word(a) >> 20) & 0x7ff);
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
PAIS_HD double det_sin(double x)
{
    if (x != x || x - x != 0.0) return x - x; // NaN / inf
    double y0, y1;
    int n = det_rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return det_ksin(y0, y1, 1);
    case 1: return det_kcos(y0, y1);
    case 2: return -det_ksin(y0, y1, 1);
    default: return -det_kcos(y0, y1);
    }
}
PAIS_HD double det_cos(double x)
{
    if (x != x || x - x != 0.0) return x - x;
    double y0, y1;
    int n = det_rem_pio2(x, &y0, &y1);
    switch (n) {
    case 0: return det_kcos(y0, y1);
    case 1: return -det_ksin(y0, y1, 1);
    case 2: return -det_kcos(y0, y1);
    default: return det_ksin(y0, y1, 1);
    }
}

} // namespace pais
