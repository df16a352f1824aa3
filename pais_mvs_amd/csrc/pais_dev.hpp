// pais_dev.hpp -- lane-local arithmetic of the PAIS-MVS hot path.
//
// Everything here is `PAIS_HD` (host+device inline) and free of wave-level
// intrinsics, so the same code is (a) inlined into the gfx950 kernels in
// pais_kernels.hip and (b) compiled for the host by tests/host_dev_shim.cpp
// where it is unit-tested against the oracle without a GPU.  It is NOT a CPU
// fallback of the product: no product entry point calls the host build.
//
// Reference citations are relative to /root/reference/TMVS/.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PAIS_HD __host__ __device__ inline
#else
#define PAIS_HD inline
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "pais_detmath.hpp"

namespace pais {

// ---------------------------------------------------------------- RNG ------
// Counter-based replacement of rand()/srand(time) (pso/psosolver.cpp:60-68):
// draw k of PSO run `run` of the candidate keyed `key`.  u = r / 2147483647.
PAIS_HD uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
PAIS_HD uint64_t stream_base(uint64_t seed, uint64_t key) { return sm64(seed ^ sm64(key)); }
PAIS_HD uint32_t rand31_from(uint64_t base, uint32_t run, uint32_t k)
{
    return (uint32_t)(sm64(base + ((((uint64_t)run) << 32) | (uint64_t)k)) >> 33);
}
PAIS_HD uint32_t rand31(uint64_t seed, uint64_t key, uint32_t run, uint32_t k)
{
    return rand31_from(stream_base(seed, key), run, k);
}
PAIS_HD double uniform_from(uint64_t base, uint32_t run, uint32_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // r / 2147483647 without the division sequence (four uniforms per particle and iteration sit on the PSO chain): q0 = r * y with
    // y = RN(1 / d), one fused correction step q1 = q0 + (r - q0 * d) * y.  Correctly rounded -- i.e. the bits of the division --
    // for EVERY r in [0, 2^31): checked exhaustively (tests/test_detmath_and_devmath.py::test_uniform_without_division_is_the_division).
    const double x = (double)rand31_from(base, run, k), d = 2147483647.0, y = 1.0 / 2147483647.0;
    const double q0 = x * y;
    return fma(fma(-q0, d, x), y, q0);
#else
    return ((double)rand31_from(base, run, k)) / ((double)2147483647);
#endif
}
PAIS_HD uint64_t child_key(uint64_t parentKey, int cam, int cx, int cy)
{
    uint64_t c = (((uint64_t)(uint32_t)cam) << 48) ^ (((uint64_t)(uint32_t)cx & 0xFFFFFFu) << 24) ^
                 ((uint64_t)(uint32_t)cy & 0xFFFFFFu);
    return sm64(sm64(parentKey) ^ (c + 0xD1B54A32D192ED03ULL));
}

// ---------------------------------------------------------------- misc -----
// cvRound (round-half-even); call sites patch.cpp:571,986,1037, mvs.cpp:860
PAIS_HD int cv_round(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_rn(v);
#else
    return (int)lrint(v);
#endif
}
PAIS_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PAIS_HD double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// utility.h:25-29 / :17-22 (sin/cos: reproducible fdlibm statements, pais_detmath.hpp)
PAIS_HD void spherical2normal(double theta, double phi, double *n)
{
    double st = det_sin(theta), ct = det_cos(theta), sp = det_sin(phi), cp = det_cos(phi);
    n[0] = st * cp;
    n[1] = st * sp;
    n[2] = ct;
}
PAIS_HD void normal2spherical(const double *n, double *s)
{
    s[0] = acos(n[2]);
    s[1] = atan2(n[1], n[0]);
}

// Camera::project without the inImage test (camera.cpp:141-157)
PAIS_HD void project_raw(const double *R, const double *T, const double *focal, const double *pp,
                         double lodScale, const double *X, double *out)
{
    double x0 = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + T[0];
    double x1 = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + T[1];
    double x2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + T[2];
    double u = focal[0] * (x0 / x2);
    double v = focal[1] * (x1 / x2);
    u += pp[0];
    v += pp[1];
    out[0] = u * lodScale;
    out[1] = v * lodScale;
}
// Camera::inImage(Vec2d, LOD) (camera.h:116-131); LOD <= maxLOD checked by caller
PAIS_HD bool in_image_d(const double *p, int w, int h)
{
    if (isnan(p[0]) || isnan(p[1])) return false;
    return !(p[0] < 0 || p[0] >= w || p[1] < 0 || p[1] >= h);
}

// 3x3 inverse the way cv::invert does it for the 3x3 double case
// (adjugate * 1/det, zeros when det == 0); call site patch.cpp:314
PAIS_HD void inv3(const double *m, double *o)
{
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c01 = m[3] * m[8] - m[5] * m[6];
    double c02 = m[3] * m[7] - m[4] * m[6];
    double d = m[0] * c00 - m[1] * c01 + m[2] * c02;
    if (d != 0.) {
        d = 1. / d;
        o[0] = c00 * d;
        o[1] = (m[2] * m[7] - m[1] * m[8]) * d;
        o[2] = (m[1] * m[5] - m[2] * m[4]) * d;
        o[3] = (m[5] * m[6] - m[3] * m[8]) * d;
        o[4] = (m[0] * m[8] - m[2] * m[6]) * d;
        o[5] = (m[2] * m[3] - m[0] * m[5]) * d;
        o[6] = (m[3] * m[7] - m[4] * m[6]) * d;
        o[7] = (m[1] * m[6] - m[0] * m[7]) * d;
        o[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    } else {
        for (int i = 0; i < 9; ++i) o[i] = 0;
    }
}

// M = d*L*KR - L*KT*n^T with L = diag(s, s, 1)   (patch.cpp:308-314,328)
PAIS_HD void plane_matrix(double d, double s, const double *KR, const double *KT, const double *n, double *M)
{
    double lkt0 = s * KT[0], lkt1 = s * KT[1], lkt2 = 1.0 * KT[2];
    M[0] = (s * KR[0]) * d - lkt0 * n[0];
    M[1] = (s * KR[1]) * d - lkt0 * n[1];
    M[2] = (s * KR[2]) * d - lkt0 * n[2];
    M[3] = (s * KR[3]) * d - lkt1 * n[0];
    M[4] = (s * KR[4]) * d - lkt1 * n[1];
    M[5] = (s * KR[5]) * d - lkt1 * n[2];
    M[6] = (1.0 * KR[6]) * d - lkt2 * n[0];
    M[7] = (1.0 * KR[7]) * d - lkt2 * n[1];
    M[8] = (1.0 * KR[8]) * d - lkt2 * n[2];
}
PAIS_HD void mul33(const double *a, const double *b, double *o)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// One bilinear sample with the reference's expression shape (patch.cpp:1005-1017,
// 365-377).  Caller has already established px, py >= 0 and in range.
PAIS_HD double bilinear(const uint8_t *img, int stride, double ix, double iy)
{
    int px0 = (int)ix, py0 = (int)iy;
    int px1 = px0 + 1, py1 = py0 + 1;
    const uint8_t *r0 = img + (size_t)py0 * stride + px0;
    const uint8_t *r1 = r0 + stride;
    double i00 = (double)r0[0], i10 = (double)r0[1], i01 = (double)r1[0], i11 = (double)r1[1];
    double ax = px1 - ix, bx = ix - px0, ay = py1 - iy, by = iy - py0;
    return i00 * ax * ay + i10 * bx * ay + i01 * ax * by + i11 * bx * by;
}

// ------------------------------------------------- fitEllipse region ratio --
// Least squares via one-sided Jacobi SVD, all sizes static so everything stays
// in registers on the GPU.  Pseudo-inverse threshold: 2*DBL_EPSILON*sum(w), as
// cvSolve(..., CV_SVD) uses.
#ifndef PAIS_JACOBI_SWEEPS
#define PAIS_JACOBI_SWEEPS 60
#endif
template <int N, int M>
PAIS_HD void jacobi_lstsq(double (&A)[N][M], const double (&b)[N], double (&x)[M])
{
    double V[M][M];
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < PAIS_JACOBI_SWEEPS; ++sweep) {
        bool changed = false;
#pragma unroll
        for (int p = 0; p < M - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < M; ++q) {
                double a = 0, bb = 0, g = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    a += A[k][p] * A[k][p];
                    bb += A[k][q] * A[k][q];
                    g += A[k][p] * A[k][q];
                }
                if (fabs(g) <= DBL_EPSILON * sqrt(a * bb) || g == 0.0) continue;
                changed = true;
                double zeta = (bb - a) / (2.0 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    double up = A[k][p], uq = A[k][q];
                    A[k][p] = c * up - s * uq;
                    A[k][q] = s * up + c * uq;
                }
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq;
                    V[k][q] = s * vp + c * vq;
                }
            }
        }
        if (!changed) break;
    }
    double w2[M], wsum = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double sq = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) sq += A[k][j] * A[k][j];
        w2[j] = sq;
        wsum += sqrt(sq);
    }
    double thr = wsum * (DBL_EPSILON * 2);
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (!(sqrt(w2[j]) > thr)) continue;
        double ub = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) ub += A[k][j] * b[k];
        const double wj = sqrt(w2[j]);
        double coef = ub / (wj * wj); // U_j = Ut_j / w_j: (Ut_j . b) / w_j^2 with the singular value itself, as the oracle
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] += V[i][j] * coef;
    }
}

// Patch::getHomographyRegionRatio (patch.cpp:269-288) with cv::fitEllipse
// (OpenCV 2.4 cvFitEllipse2, Weiss' algebraic fit) on the 8 warped window points.
PAIS_HD double region_ratio(double ptx, double pty, int r, const double *H)
{
    const double xs[8] = {ptx - r, ptx - r, ptx + r, ptx + r, ptx - r, ptx, ptx + r, ptx};
    const double ys[8] = {pty - r, pty + r, pty + r, pty - r, pty, pty + r, pty, pty - r};
    float px[8], py[8];
    float cx = 0, cy = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double w = H[6] * xs[i] + H[7] * ys[i] + H[8];
        px[i] = (float)((H[0] * xs[i] + H[1] * ys[i] + H[2]) / w);
        py[i] = (float)((H[3] * xs[i] + H[4] * ys[i] + H[5]) / w);
        cx += px[i];
        cy += py[i];
    }
    cx /= 8;
    cy /= 8;
    double gfp[5], rp[5];
    {
        double A[8][5], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = px[i] - cx, y = py[i] - cy;
            b[i] = 10000.0;
            A[i][0] = -(double)x * x;
            A[i][1] = -(double)y * y;
            A[i][2] = -(double)x * y;
            A[i][3] = x;
            A[i][4] = y;
        }
        jacobi_lstsq<8, 5>(A, b, gfp);
    }
    {
        double A[2][2] = {{2 * gfp[0], gfp[2]}, {gfp[2], 2 * gfp[1]}};
        double b[2] = {gfp[3], gfp[4]};
        double x2[2];
        jacobi_lstsq<2, 2>(A, b, x2);
        rp[0] = x2[0];
        rp[1] = x2[1];
    }
    {
        double A[8][3], b[8], x3[3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = px[i] - cx, y = py[i] - cy;
            b[i] = 1.0;
            A[i][0] = (x - rp[0]) * (x - rp[0]);
            A[i][1] = (y - rp[1]) * (y - rp[1]);
            A[i][2] = (x - rp[0]) * (y - rp[1]);
        }
        jacobi_lstsq<8, 3>(A, b, x3);
        gfp[0] = x3[0];
        gfp[1] = x3[1];
        gfp[2] = x3[2];
    }
    const double min_eps = 1e-6;
    rp[4] = -0.5 * atan2(gfp[2], gfp[1] - gfp[0]);
    double t = sin(-2.0 * rp[4]);
    if (fabs(t) > fabs(gfp[2]) * min_eps)
        t = gfp[2] / t;
    else
        t = gfp[1] - gfp[0];
    rp[2] = fabs(gfp[0] + gfp[1] - t);
    if (rp[2] > min_eps) rp[2] = sqrt(2.0 / rp[2]);
    rp[3] = fabs(gfp[0] + gfp[1] + t);
    if (rp[3] > min_eps) rp[3] = sqrt(2.0 / rp[3]);
    float bw = (float)(rp[2] * 2), bh = (float)(rp[3] * 2);
    float mn = bw < bh ? bw : bh, mx = bw > bh ? bw : bh;
    return (double)(mn / mx);
}

// --------------------------------------------------------------- PSO -------
// One particle's share of PsoSolver::moveParticles with GLN-PSO enabled
// (pso/psosolver.cpp:220-265): getLocalBest (:151-191), setNearNeighborBest
// (:193-218), velocity/position update and clamp.  The arrays are the whole
// swarm; only row i of pos/vec/nBest is written.
// u[4] = the four uniforms of this particle in the order p, g, l, n (:232-237).
PAIS_HD void pso_move_particle(int i, int N, int localK, double iw, const double *u,
                               double (*pos)[3], double (*vec)[3], const double (*pBest)[3],
                               double (*nBest)[3], const double *fit, const double *pBestFit,
                               const double *gBest, const double *rangeL, const double *rangeU)
{
    const double pw = 1.2, gw = 1.5, lw = 1.0, nw = 1.0; // psosolver.h:110
    double pVecW = pw * u[0], gVecW = gw * u[1], lVecW = lw * u[2], nVecW = nw * u[3];

    // getLocalBest: localK nearest pBests by (squared distance, index) -- the
    // reference's std::sort is an insertion sort at these sizes, i.e. stable.
    const double pp0 = pBest[i][0], pp1 = pBest[i][1], pp2 = pBest[i][2];
    uint64_t taken0 = 0, taken1 = 0;
    double minFitness = DBL_MAX;
    int lIdx = i;
    for (int k = 0; k < localK; ++k) {
        double bd = 0;
        int bj = -1;
        for (int j = 0; j < N; ++j) {
            bool tk = (j < 64) ? ((taken0 >> j) & 1) : ((taken1 >> (j - 64)) & 1);
            if (tk) continue;
            double dist;
            if (j == i) {
                dist = DBL_MAX;
            } else {
                double d0 = pp0 - pBest[j][0], d1 = pp1 - pBest[j][1], d2 = pp2 - pBest[j][2];
                dist = 0;
                dist += d0 * d0;
                dist += d1 * d1;
                dist += d2 * d2;
            }
            if (bj < 0 || dist < bd) {
                bd = dist;
                bj = j;
            }
        }
        if (bj < 0) break;
        if (bj < 64) taken0 |= (1ULL << bj); else taken1 |= (1ULL << (bj - 64));
        if (pBestFit[bj] < minFitness) {
            minFitness = pBestFit[bj];
            lIdx = bj;
        }
    }
    // setNearNeighborBest
    const double fitness = fit[i];
    for (int d = 0; d < 3; ++d) {
        double maxFDR = -DBL_MAX;
        double nb = nBest[i][d];
        const double pd = pos[i][d];
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            double FDR = (fitness - pBestFit[j]) / fabs(pd - pBest[j][d]);
            if (FDR > maxFDR) {
                maxFDR = FDR;
                nb = pBest[j][d];
            }
        }
        nBest[i][d] = nb;
    }
    for (int d = 0; d < 3; ++d) {
        double p = pos[i][d];
        double v = iw * vec[i][d] + pVecW * (pBest[i][d] - p) + gVecW * (gBest[d] - p) +
                   lVecW * (pBest[lIdx][d] - p) + nVecW * (nBest[i][d] - p);
        vec[i][d] = v;
        p += v;
        if (p > rangeU[d]) p = rangeU[d];
        if (p < rangeL[d]) p = rangeL[d];
        pos[i][d] = p;
    }
}

} // namespace pais
