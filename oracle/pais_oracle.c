/*
 * pais_oracle.c -- CPU restatement of the pais-mvs hot path (see pais_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product.
 *
 * Style: each function follows the cited reference lines statement by
 * statement (same loop order, same comparison operators, same association of
 * floating-point expressions).  Where the reference calls into OpenCV 2.4.2
 * (not present in /root/reference; pinned by TMVS.vcxproj:148 / README.md:38)
 * the published OpenCV algorithm is restated and marked "OpenCV:".
 *
 * Build: gcc -O2 -std=c99 -fPIC -shared -fopenmp -ffp-contract=off (Makefile).
 */
#include "pais_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "po_detmath.h"

/* Elementary functions: the platform libm by default (what the reference does);
 * the reproducible fdlibm statements when the scene asks for them (see
 * po_detmath.h and DESIGN.md 5.3). */
static inline double m_exp(const po_scene *s, double x) { return s->detMath ? po_det_exp(x) : exp(x); }
static inline double m_sin(const po_scene *s, double x) { return s->detMath ? po_det_sin(x) : sin(x); }
static inline double m_cos(const po_scene *s, double x) { return s->detMath ? po_det_cos(x) : cos(x); }
double po_exp_det(double x) { return po_det_exp(x); }
double po_exp_poly(double x) { return po_det_exp_poly(x); }
double po_sin_det(double x) { return po_det_sin(x); }
double po_cos_det(double x) { return po_det_cos(x); }

/* Sum of 64 partials in the order of a wave64 xor-butterfly (v += shfl_xor(v, m),
 * m = 32,16,...,1): the reduction tree of the HIP kernels (po_scene.treeSum). */
static double tree64(const double *p)
{
    double a[64], b[64];
    for (int l = 0; l < 64; ++l) a[l] = p[l];
    for (int m = 32; m >= 1; m >>= 1) {
        for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ m];
        for (int l = 0; l < 64; ++l) a[l] = b[l];
    }
    return a[0];
}

/* OpenCV: cvRound == lrint under the default rounding mode (round-half-even).
 * Call sites: patch.cpp:571-572,651,986,1037; mvs.cpp:860. */
static inline int cv_round(double v) { return (int)lrint(v); }
/* OpenCV: cvCeil (cellmap.cpp:7-8) */
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

/* ------------------------------------------------------------------------ */
/* deterministic uniform stream                                              */
/* ------------------------------------------------------------------------ */
static inline uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* k-th 31-bit draw of PSO run `run` of the candidate with key `key`.
 * Replaces rand() after srand(time+tid) (psosolver.cpp:60-68): the reference
 * is non-deterministic; u = r / 2147483647.0 as glibc RAND_MAX. */
uint32_t po_rand31(uint64_t seed, uint64_t key, uint32_t run, uint32_t k)
{
    uint64_t a = sm64(seed ^ sm64(key));
    uint64_t b = sm64(a + ((((uint64_t)run) << 32) | (uint64_t)k));
    return (uint32_t)(b >> 33);
}

uint64_t po_child_key(uint64_t parentKey, int cam, int cx, int cy)
{
    uint64_t c = (((uint64_t)(uint32_t)cam) << 48) ^ (((uint64_t)(uint32_t)cx & 0xFFFFFFu) << 24) ^
                 ((uint64_t)(uint32_t)cy & 0xFFFFFFu);
    return sm64(sm64(parentKey) ^ (c + 0xD1B54A32D192ED03ULL));
}

/* ------------------------------------------------------------------------ */
/* config                                                                    */
/* ------------------------------------------------------------------------ */
/* TMVS.cpp:26-52 */
void po_config_defaults(po_config *c)
{
    memset(c, 0, sizeof(*c));
    c->cellSize = 4;
    c->patchRadius = 15;
    c->reduceNormalRange = 2;
    c->adaptiveDistanceEnable = 1;
    c->adaptiveDifferenceEnable = 1;
    c->adaptiveGradientEnable = 0;
    c->distWeighting = c->patchRadius / 3.0;
    c->diffWeighting = 128 * 128;
    c->gradientWeighting = 10.0;
    c->minCamNum = 3;
    c->textureVariation = 36;
    c->visibleCorrelation = 0.7;
    c->minCorrelation = 0.7;
    c->maxFitness = 10.0;
    c->minLOD = 0;
    c->maxLOD = 15;
    c->lodRatio = 0.8;
    c->maxCellPatchNum = 3;
    c->neighborRadius = 0.005;
    c->neighborRadiusScalar = 0.0025;
    c->minRegionRatio = 0.55;
    c->depthRangeScalar = 1;
    c->particleNum = 5;
    c->maxIteration = 10;
    c->expansionStrategy = 0;
    c->patchSize = (c->patchRadius << 1) + 1; /* mvs.cpp:67 */
}

/* README.md:110-207 (the documented config.txt) on top of the defaults */
void po_config_readme(po_config *c)
{
    po_config_defaults(c);
    c->patchRadius = 15;
    c->reduceNormalRange = 2;
    c->adaptiveDistanceEnable = 1;
    c->distWeighting = 5;
    c->adaptiveDifferenceEnable = 1;
    c->diffWeighting = 16384;
    c->adaptiveGradientEnable = 0;
    c->gradientWeighting = 10.0;
    c->visibleCorrelation = 0.7;
    c->depthRangeScalar = 8;
    c->particleNum = 15;
    c->maxIteration = 30;
    c->cellSize = 2;
    c->maxCellPatchNum = 3;
    c->expansionStrategy = 0;
    c->textureVariation = 36;
    c->minLOD = 0;
    c->maxLOD = 15;
    c->lodRatio = 0.8;
    c->minCamNum = 3;
    c->minCorrelation = 0.9;
    c->minRegionRatio = 0.15;
    c->maxFitness = 10.0;
    c->neighborRadiusScalar = 0.01;
    c->patchSize = (c->patchRadius << 1) + 1;
}

/* mvs.cpp:97-114 */
void po_init_gauss(const po_config *cfg, double *out)
{
    const int patchSize = cfg->patchSize, patchRadius = cfg->patchRadius;
    double sigma = cfg->distWeighting;
    double s2 = 1.0 / (2.0 * sigma * sigma);
    double s = 1.0 / (2.0 * M_PI * sigma * sigma);
    double e, g;
    for (int x = 0; x < patchSize; ++x) {
        for (int y = 0; y < patchSize; ++y) {
            e = -(pow((double)(x - patchRadius), 2) + pow((double)(y - patchRadius), 2)) * s2;
            g = s * exp(e);
            out[x * patchSize + y] = g; /* .at<double>(x,y): row x, col y */
        }
    }
    /* OpenCV: cv::sum accumulates row-major; patchDistWeight / n[0] is the
     * MatExpr a*(1/s) (operator/(Mat,double) -> scale by 1./s). */
    double n = 0;
    for (int i = 0; i < patchSize * patchSize; ++i) n += out[i];
    double inv = 1.0 / n;
    for (int i = 0; i < patchSize * patchSize; ++i) out[i] = out[i] * inv;
}

/* ------------------------------------------------------------------------ */
/* camera                                                                    */
/* ------------------------------------------------------------------------ */
/* OpenCV: 3x3 * 3x3 gemm, s accumulated over k = 0..2 starting from 0 */
static void mat33_mul(const double *a, const double *b, double *out)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            t[i * 3 + j] = s;
        }
    memcpy(out, t, sizeof(t));
}
static void mat33_vec(const double *a, const double *v, double *out)
{
    double t[3];
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * v[k];
        t[i] = s;
    }
    memcpy(out, t, sizeof(t));
}

/* camera.cpp:6-35 (quaternion -> R) and camera.cpp:108-133 */
void po_camera_init(po_camera *cam, const double focal[2], const double pp[2],
                    const double q[4], const double center[3])
{
    const double qq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double qw, qx, qy, qz;
    if (qq > 0) {
        qw = q[0] / qq; qx = q[1] / qq; qy = q[2] / qq; qz = q[3] / qq;
    } else {
        qw = 1; qx = qy = qz = 0;
    }
    double *R = cam->R;
    R[0] = qw * qw + qx * qx - qz * qz - qy * qy;
    R[1] = 2 * qx * qy - 2 * qz * qw;
    R[2] = 2 * qy * qw + 2 * qz * qx;
    R[3] = 2 * qx * qy + 2 * qw * qz;
    R[4] = qy * qy + qw * qw - qz * qz - qx * qx;
    R[5] = 2 * qz * qy - 2 * qx * qw;
    R[6] = 2 * qx * qz - 2 * qy * qw;
    R[7] = 2 * qy * qz + 2 * qw * qx;
    R[8] = qz * qz + qw * qw - qy * qy - qx * qx;

    cam->focal[0] = focal[0]; cam->focal[1] = focal[1];
    cam->pp[0] = pp[0]; cam->pp[1] = pp[1];
    for (int i = 0; i < 3; ++i) cam->C[i] = center[i];

    double K[9] = {focal[0], 0.0, pp[0], 0, focal[1], pp[1], 0, 0, 1};
    /* translation = -rotation * Mat(center): gemm with alpha = -1 */
    double rc[3];
    mat33_vec(R, center, rc);
    for (int i = 0; i < 3; ++i) cam->T[i] = rc[i] * -1.0;
    mat33_mul(K, R, cam->KR);
    mat33_vec(K, cam->T, cam->KT);
    /* dir = R^T * (0,0,1) */
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        s += R[0 * 3 + i] * 0.0;
        s += R[1 * 3 + i] * 0.0;
        s += R[2 * 3 + i] * 1.0;
        cam->optN[i] = s;
    }
}

/* camera.cpp:63-64 */
int po_camera_max_lod(int width, int height, double lodRatio, int cfgMaxLOD)
{
    int mx = width > height ? width : height;
    int maxLOD = (int)(log((double)mx) / log(1.0 / lodRatio));
    return maxLOD < cfgMaxLOD ? maxLOD : cfgMaxLOD;
}

po_scene *po_scene_create(const po_config *cfg, int numCams, const po_camera *cams, uint64_t seed)
{
    po_scene *s = (po_scene *)calloc(1, sizeof(po_scene));
    s->cfg = *cfg;
    s->cfg.patchSize = (cfg->patchRadius << 1) + 1;
    s->numCams = numCams;
    s->cams = (po_camera *)malloc(sizeof(po_camera) * (size_t)numCams);
    memcpy(s->cams, cams, sizeof(po_camera) * (size_t)numCams);
    s->gauss = (double *)malloc(sizeof(double) * (size_t)s->cfg.patchSize * s->cfg.patchSize);
    po_init_gauss(&s->cfg, s->gauss);
    for (int l = 0; l < PO_MAX_LEVELS; ++l) s->lodScale[l] = pow(s->cfg.lodRatio, l);
    s->seed = seed;
    s->ompParticles = 0;
    return s;
}

void po_scene_destroy(po_scene *s)
{
    if (!s) return;
    free(s->cams);
    free(s->gauss);
    free(s);
}

/* ------------------------------------------------------------------------ */
/* geometry                                                                  */
/* ------------------------------------------------------------------------ */
/* utility.h:25-29 */
void po_spherical2normal(const double in[2], double out[3])
{
    out[0] = sin(in[0]) * cos(in[1]);
    out[1] = sin(in[0]) * sin(in[1]);
    out[2] = cos(in[0]);
}
/* utility.h:17-22 */
void po_normal2spherical(const double in[3], double out[2])
{
    out[0] = acos(in[2]);
    out[1] = atan2(in[1], in[0]);
}

static void s2n(const po_scene *s, const double in[2], double out[3])
{
    out[0] = m_sin(s, in[0]) * m_cos(s, in[1]);
    out[1] = m_sin(s, in[0]) * m_sin(s, in[1]);
    out[2] = m_cos(s, in[0]);
}

static inline double dot3(const double *a, const double *b)
{
    /* OpenCV: Matx::ddot, s += a[i]*b[i] sequentially */
    double s = 0;
    for (int i = 0; i < 3; ++i) s += a[i] * b[i];
    return s;
}
static inline double norm3(const double *a)
{
    /* OpenCV: norm(Vec) = sqrt(normL2Sqr), sequential for n < 4 */
    double s = 0;
    for (int i = 0; i < 3; ++i) s += a[i] * a[i];
    return sqrt(s);
}

/* camera.h:116-131 */
static int in_image_d(const po_camera *cam, const double p[2], int LOD)
{
    if (LOD > cam->maxLOD) return 0;
    if (isnan(p[0]) || isnan(p[1])) return 0;
    if (p[0] < 0 || p[0] >= cam->width[LOD] || p[1] < 0 || p[1] >= cam->height[LOD]) return 0;
    return 1;
}
/* camera.h:133-148 */
static int in_image_i(const po_camera *cam, int x, int y, int LOD)
{
    if (LOD > cam->maxLOD) return 0;
    if (x < 0 || x >= cam->width[LOD] || y < 0 || y >= cam->height[LOD]) return 0;
    return 1;
}

/* camera.cpp:138-160 (applyDistortion == false, its default) */
int po_project(const po_scene *s, int camI, const double X[3], double out[2], int LOD)
{
    const po_camera *cam = &s->cams[camI];
    /* OpenCV: X2 = rotation * X + translation  -> gemm, (sum_k) + T */
    double X2[3];
    for (int i = 0; i < 3; ++i) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += cam->R[i * 3 + k] * X[k];
        X2[i] = a + cam->T[i];
    }
    out[0] = cam->focal[0] * (X2[0] / X2[2]);
    out[1] = cam->focal[1] * (X2[1] / X2[2]);
    out[0] += cam->pp[0];
    out[1] += cam->pp[1];
    double sc = s->lodScale[LOD]; /* pow(mvs.lodRatio, LOD) */
    out[0] *= sc;
    out[1] *= sc;
    return in_image_d(cam, out, LOD);
}

/* OpenCV 2.4 cv::invert, DECOMP_LU, 3x3 double special case: adjugate / det3;
 * singular (det == 0) -> result is all zeros.  Call site patch.cpp:314. */
int po_inv3(const double m[9], double o[9])
{
#define M_(r, c) m[(r) * 3 + (c)]
    double d = M_(0, 0) * (M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(2, 1)) -
               M_(0, 1) * (M_(1, 0) * M_(2, 2) - M_(1, 2) * M_(2, 0)) +
               M_(0, 2) * (M_(1, 0) * M_(2, 1) - M_(1, 1) * M_(2, 0));
    if (d != 0.) {
        double t[9];
        d = 1. / d;
        t[0] = (M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(2, 1)) * d;
        t[1] = (M_(0, 2) * M_(2, 1) - M_(0, 1) * M_(2, 2)) * d;
        t[2] = (M_(0, 1) * M_(1, 2) - M_(0, 2) * M_(1, 1)) * d;
        t[3] = (M_(1, 2) * M_(2, 0) - M_(1, 0) * M_(2, 2)) * d;
        t[4] = (M_(0, 0) * M_(2, 2) - M_(0, 2) * M_(2, 0)) * d;
        t[5] = (M_(0, 2) * M_(1, 0) - M_(0, 0) * M_(1, 2)) * d;
        t[6] = (M_(1, 0) * M_(2, 1) - M_(1, 1) * M_(2, 0)) * d;
        t[7] = (M_(0, 1) * M_(2, 0) - M_(0, 0) * M_(2, 1)) * d;
        t[8] = (M_(0, 0) * M_(1, 1) - M_(0, 1) * M_(1, 0)) * d;
        memcpy(o, t, sizeof(t));
        return 1;
    }
    for (int i = 0; i < 9; ++i) o[i] = 0;
    return 0;
#undef M_
}

/* d*LODM*KR - LODM*KT*n^T  (patch.cpp:314,328).
 * OpenCV: d*LODM*KR is gemm(LODM, KR, alpha=d): (sum_k L_ik KR_kj) * d with L
 * diagonal; LODM*KT*n^T is gemm(gemm(LODM,KT), n, GEMM_2_T). */
static void plane_matrix(double d, double sc, const double *KR, const double *KT,
                         const double *n, double *M)
{
    double L[3] = {sc, sc, 1.0};
    for (int i = 0; i < 3; ++i) {
        double lkt = L[i] * KT[i];
        for (int j = 0; j < 3; ++j) {
            double a = (L[i] * KR[i * 3 + j]) * d;
            double b = lkt * n[j];
            M[i * 3 + j] = a - b;
        }
    }
}

/* patch.cpp:290-330 */
void po_homographies(const po_scene *s, const po_patch *p, const double center[3],
                     const double normal[3], double *H)
{
    const po_camera *refCam = &s->cams[p->refCamIdx];
    const double d = -dot3(center, normal);
    const int camNum = p->numCam;
    const double sc = s->lodScale[p->LOD];

    double Mref[9], invH[9];
    plane_matrix(d, sc, refCam->KR, refCam->KT, normal, Mref);
    po_inv3(Mref, invH);
    for (int i = 0; i < camNum; i++) {
        double *Hi = H + 9 * i;
        if (p->camIdx[i] == p->refCamIdx) {
            Hi[0] = 1; Hi[1] = 0; Hi[2] = 0;
            Hi[3] = 0; Hi[4] = 1; Hi[5] = 0;
            Hi[6] = 0; Hi[7] = 0; Hi[8] = 1;
            continue;
        }
        const po_camera *cam = &s->cams[p->camIdx[i]];
        double M[9];
        plane_matrix(d, sc, cam->KR, cam->KT, normal, M);
        mat33_mul(M, invH, Hi);
    }
}

/* ------------------------------------------------------------------------ */
/* OpenCV 2.4 fitEllipse (imgproc/shapedescr.cpp, cvFitEllipse2, "New        */
/* fitellipse algorithm, contributed by Dr. Daniel Weiss") -- restated.       */
/* LS solves use the SVD pseudo-inverse (cvSolve(..., CV_SVD)); here: one-    */
/* sided Jacobi SVD, singular values <= 2*DBL_EPSILON*sum(w) treated as 0.    */
/* ------------------------------------------------------------------------ */
static void svd_solve(int n, int m, const double *A, const double *b, double *x)
{
    /* one-sided Jacobi (Hestenes) on the columns of a copy of A (n x m, m<=5) */
    double U[8 * 5 + 64];
    double V[25];
    double w[5];
    double *Ut = (n * m <= 8 * 5 + 64) ? U : (double *)malloc(sizeof(double) * (size_t)(n * m));
    for (int i = 0; i < n * m; ++i) Ut[i] = A[i];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) V[i * m + j] = (i == j) ? 1.0 : 0.0;

    for (int sweep = 0; sweep < 60; ++sweep) {
        int changed = 0;
        for (int p = 0; p < m - 1; ++p) {
            for (int q = p + 1; q < m; ++q) {
                double a = 0, bb = 0, g = 0;
                for (int k = 0; k < n; ++k) {
                    double up = Ut[k * m + p], uq = Ut[k * m + q];
                    a += up * up; bb += uq * uq; g += up * uq;
                }
                if (fabs(g) <= DBL_EPSILON * sqrt(a * bb) || g == 0.0) continue;
                changed = 1;
                double zeta = (bb - a) / (2.0 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; ++k) {
                    double up = Ut[k * m + p], uq = Ut[k * m + q];
                    Ut[k * m + p] = c * up - sn * uq;
                    Ut[k * m + q] = sn * up + c * uq;
                }
                for (int k = 0; k < m; ++k) {
                    double vp = V[k * m + p], vq = V[k * m + q];
                    V[k * m + p] = c * vp - sn * vq;
                    V[k * m + q] = sn * vp + c * vq;
                }
            }
        }
        if (!changed) break;
    }
    double wsum = 0;
    for (int j = 0; j < m; ++j) {
        double sq = 0;
        for (int k = 0; k < n; ++k) sq += Ut[k * m + j] * Ut[k * m + j];
        w[j] = sqrt(sq);
        wsum += w[j];
    }
    double threshold = wsum * (DBL_EPSILON * 2);
    /* x = V * diag(1/w) * U^T b ; U_j = Ut_j / w_j  => coefficient (Ut_j . b)/w_j^2 */
    for (int i = 0; i < m; ++i) x[i] = 0;
    for (int j = 0; j < m; ++j) {
        if (!(w[j] > threshold)) continue;
        double ub = 0;
        for (int k = 0; k < n; ++k) ub += Ut[k * m + j] * b[k];
        double coef = ub / (w[j] * w[j]);
        for (int i = 0; i < m; ++i) x[i] += V[i * m + j] * coef;
    }
    if (Ut != U) free(Ut);
}

void po_fit_ellipse(int n, const float *xy, float *ocx, float *ocy, float *ow, float *oh, float *oangle)
{
    float cx = 0, cy = 0;
    double gfp[5], rp[5], t;
    const double min_eps = 1e-6;
    double Ad[64 * 5] = {0}, bd[64] = {0};
    if (n > 64) n = 64;

    for (int i = 0; i < n; i++) { cx += xy[2 * i]; cy += xy[2 * i + 1]; }
    cx /= n;
    cy /= n;

    for (int i = 0; i < n; i++) {
        float px = xy[2 * i], py = xy[2 * i + 1];
        px -= cx;
        py -= cy;
        bd[i] = 10000.0;
        Ad[i * 5] = -(double)px * px;
        Ad[i * 5 + 1] = -(double)py * py;
        Ad[i * 5 + 2] = -(double)px * py;
        Ad[i * 5 + 3] = px;
        Ad[i * 5 + 4] = py;
    }
    svd_solve(n, 5, Ad, bd, gfp);

    Ad[0] = 2 * gfp[0];
    Ad[1] = Ad[2] = gfp[2];
    Ad[3] = 2 * gfp[1];
    bd[0] = gfp[3];
    bd[1] = gfp[4];
    svd_solve(2, 2, Ad, bd, rp);

    for (int i = 0; i < n; i++) {
        float px = xy[2 * i], py = xy[2 * i + 1];
        px -= cx;
        py -= cy;
        bd[i] = 1.0;
        Ad[i * 3] = (px - rp[0]) * (px - rp[0]);
        Ad[i * 3 + 1] = (py - rp[1]) * (py - rp[1]);
        Ad[i * 3 + 2] = (px - rp[0]) * (py - rp[1]);
    }
    svd_solve(n, 3, Ad, bd, gfp);

    rp[4] = -0.5 * atan2(gfp[2], gfp[1] - gfp[0]);
    t = sin(-2.0 * rp[4]);
    if (fabs(t) > fabs(gfp[2]) * min_eps)
        t = gfp[2] / t;
    else
        t = gfp[1] - gfp[0];
    rp[2] = fabs(gfp[0] + gfp[1] - t);
    if (rp[2] > min_eps) rp[2] = sqrt(2.0 / rp[2]);
    rp[3] = fabs(gfp[0] + gfp[1] + t);
    if (rp[3] > min_eps) rp[3] = sqrt(2.0 / rp[3]);

    float bcx = (float)rp[0] + cx;
    float bcy = (float)rp[1] + cy;
    float bw = (float)(rp[2] * 2);
    float bh = (float)(rp[3] * 2);
    float ang = 0;
    if (bw > bh) {
        float tmp = bw; bw = bh; bh = tmp;
        ang = (float)(90 + rp[4] * 180 / M_PI);
    }
    if (ang < -180) ang += 360;
    if (ang > 360) ang -= 360;
    if (ocx) *ocx = bcx;
    if (ocy) *ocy = bcy;
    if (ow) *ow = bw;
    if (oh) *oh = bh;
    if (oangle) *oangle = ang;
}

/* patch.cpp:269-288 */
double po_region_ratio(const po_scene *s, const double pt[2], const double H[9])
{
    const int patchRadius = s->cfg.patchRadius;
    double x[] = {pt[0] - patchRadius, pt[0] - patchRadius, pt[0] + patchRadius, pt[0] + patchRadius,
                  pt[0] - patchRadius, pt[0], pt[0] + patchRadius, pt[0]};
    double y[] = {pt[1] - patchRadius, pt[1] + patchRadius, pt[1] + patchRadius, pt[1] - patchRadius,
                  pt[1], pt[1] + patchRadius, pt[1], pt[1] - patchRadius};
    float p[16];
    double w;
    for (int i = 0; i < 8; ++i) {
        w = H[6] * x[i] + H[7] * y[i] + H[8];
        p[2 * i] = (float)((H[0] * x[i] + H[1] * y[i] + H[2]) / w);
        p[2 * i + 1] = (float)((H[3] * x[i] + H[4] * y[i] + H[5]) / w);
    }
    float bw, bh;
    po_fit_ellipse(8, p, NULL, NULL, &bw, &bh, NULL);
    float mn = bw < bh ? bw : bh, mx = bw > bh ? bw : bh;
    return (double)(mn / mx);
}

typedef struct {
    const po_scene *s; const po_patch *patch; const double *H; const uint8_t *refImg; const double *edgeImg;
    int refCols, LOD, camNum;
} fit_px_ctx;

/* one window pixel of PAIS::getFitness (patch.cpp:981-1038).
 * returns 0: masked (skipped), 1: ok (*weight, *avgSad set), -1: overflow -> whole call DBL_MAX */
static int fit_pixel(const fit_px_ctx *f, double x, double y, double distW, double *weightOut, double *sadOut)
{
    const po_scene *s = f->s;
    const po_patch *patch = f->patch;
    const int camNum = f->camNum, LOD = f->LOD;
    double mean = 0, avgSad = 0;
    double w, ix, iy;
    int px[4], py[4];
    double c[PO_MAX_VIS];

    if (f->refImg[(size_t)cv_round(y) * f->refCols + cv_round(x)] == 0) return 0; /* :986 */

    for (int i = 0; i < camNum; ++i) {
        const po_camera *cam = &s->cams[patch->camIdx[i]];
        const uint8_t *img = cam->img[LOD];
        const int cols = cam->width[LOD], rows = cam->height[LOD];
        const double *Hi = f->H + 9 * i;

        if (s->literalVariant & 4) { /* control: a * x + b * y + c contracted = fma(b, y, a * x) + c */
            w = fma(Hi[7], y, Hi[6] * x) + Hi[8];
            ix = (fma(Hi[1], y, Hi[0] * x) + Hi[2]) / w;
            iy = (fma(Hi[4], y, Hi[3] * x) + Hi[5]) / w;
        } else {
            w = (Hi[6] * x + Hi[7] * y + Hi[8]);
            ix = (Hi[0] * x + Hi[1] * y + Hi[2]) / w;
            iy = (Hi[3] * x + Hi[4] * y + Hi[5]) / w;
        }

        if (ix < 2 || ix >= cols - 3 || iy < 2 || iy >= rows - 3 || w == 0) return -1; /* :999 */
        /* (int)NaN is UB in the reference; NaN passes the test above only if w
         * is NaN -- treated as overflow here. */
        if (isnan(ix) || isnan(iy)) return -1;

        px[0] = (int)ix; py[0] = (int)iy;
        px[1] = px[0] + 1; py[1] = py[0];
        px[2] = px[0]; py[2] = py[0] + 1;
        px[3] = px[0] + 1; py[3] = py[0] + 1;

        if (s->literalVariant & 4) { /* control: ((t0 + t1) + t2) + t3 with t = (img * u) * v contracted left to right */
            const double i0 = (double)img[(size_t)py[0] * cols + px[0]], i1 = (double)img[(size_t)py[1] * cols + px[1]];
            const double i2 = (double)img[(size_t)py[2] * cols + px[2]], i3 = (double)img[(size_t)py[3] * cols + px[3]];
            double acc = i0 * (px[1] - ix) * (py[2] - iy);
            acc = fma(i1 * (ix - px[0]), (py[2] - iy), acc);
            acc = fma(i2 * (px[1] - ix), (iy - py[0]), acc);
            acc = fma(i3 * (ix - px[0]), (iy - py[0]), acc);
            c[i] = acc;
        } else
        c[i] = (double)img[(size_t)py[0] * cols + px[0]] * (px[1] - ix) * (py[2] - iy) +
               (double)img[(size_t)py[1] * cols + px[1]] * (ix - px[0]) * (py[2] - iy) +
               (double)img[(size_t)py[2] * cols + px[2]] * (px[1] - ix) * (iy - py[0]) +
               (double)img[(size_t)py[3] * cols + px[3]] * (ix - px[0]) * (iy - py[0]);
        mean += c[i];
    }
    mean /= camNum;
    for (int i = 0; i < camNum; i++) avgSad += fabs(c[i] - mean);
    avgSad /= camNum;

    double weight = 1;
    if (s->cfg.adaptiveDistanceEnable) weight *= distW;
    if (s->cfg.adaptiveDifferenceEnable) weight *= m_exp(s, -avgSad * avgSad / s->cfg.diffWeighting);
    if (s->cfg.adaptiveGradientEnable)
        weight *= m_exp(s, -1.0 / (f->edgeImg[(size_t)cv_round(y) * f->refCols + cv_round(x)] * s->cfg.gradientWeighting));
    *weightOut = weight;
    *sadOut = avgSad;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* "Kernel arithmetic" v5 of the cost: the same function as   */
/* po_get_fitness's literal branch, evaluated the way the HIP kernels do      */
/* (pais_mvs_amd/csrc/pais_eval.hpp) -- differs in the last bits only:        */
/*  * every particle's centre ray*depth + C_ref projects to the same point of */
/*    the reference camera up to rounding, so the window origin is taken once */
/*    per run from ray*1 + C_ref, and with it the mask, the reference camera's*/
/*    own colour (identity homography: a bilinear sample at (x, y)) and the   */
/*    distance / gradient weights of every window pixel;                      */
/*  * homography rows with fma, ONE reciprocal per group of the other cameras */
/*    (pairs, one triple for an odd count), bilinear as three lerps,          */
/*    colours summed reference first (from 13 cameras on in TWO groups: the  */
/*    reference + the first 2*((M+4)/4) cameras, then the rest -- round 6),   */
/*    mean / SAD scaled by 1/K, polynomial exp, weight = (dist * grad) * diff;*/
/*  * pixel k = yi*S + xi goes to lane k%64 of sub-accumulator (k/64)%4; each */
/*    (sub-accumulator, lane) adds its pixels in increasing k; wave64         */
/*    butterfly per sub-accumulator, then ((a0 + a1) + a2) + a3.              */
/* ------------------------------------------------------------------------ */
#define PO_TWO_LEVEL_K 13 /* == PAIS_TWO_LEVEL_K of the HIP path (pais_eval.hpp) */
static double lerp3_u8(const uint8_t *img, int cols, int qx, int qy, double bx, double by)
{
    const uint8_t *r0 = img + (size_t)qy * cols + qx, *r1 = r0 + cols;
    /* three lerps a + f (b - a); the pixel differences are exact */
    const double t0 = fma(bx, (double)((int)r0[1] - (int)r0[0]), (double)r0[0]);
    const double t1 = fma(bx, (double)((int)r1[1] - (int)r1[0]), (double)r1[0]);
    return fma(by, t1 - t0, t0);
}

static double get_fitness_kernel(const po_scene *s, const po_patch *patch, const double pos[3])
{
    const int r = s->cfg.patchRadius, S = s->cfg.patchSize;
    const int LOD = patch->LOD;
    const po_camera *refCam = &s->cams[patch->refCamIdx];
    const int K = patch->numCam;
    const uint8_t *refImg = refCam->img[LOD];
    const double *edgeImg = refCam->edge[LOD];
    const int refCols = refCam->width[LOD], refRows = refCam->height[LOD];

    double normal[3];
    double sph[2] = {pos[0], pos[1]};
    s2n(s, sph, normal);
    if (dot3(normal, refCam->optN) > 0) return DBL_MAX; /* :939 */

    /* the window of the run (:952-962); windowPerParticle (diagnosis only, tests/test_oracle_modes.py): from the
     * particle's own centre as the reference does -- the same point up to rounding */
    double c1[3], pt[2];
    for (int i = 0; i < 3; ++i) c1[i] = patch->ray[i] * (s->windowPerParticle ? pos[2] : 1.0) + refCam->C[i];
    if (!po_project(s, patch->refCamIdx, c1, pt, LOD)) return DBL_MAX;
    if (pt[0] - r < 2 || pt[0] + r >= refCols - 3 || pt[1] - r < 2 || pt[1] + r >= refRows - 3) return DBL_MAX;
    if (!(fabs(pos[2]) > 0)) return DBL_MAX; /* centre == C_ref: the reference's own projection is 0/0 */

    double center[3];
    for (int i = 0; i < 3; ++i) center[i] = patch->ray[i] * pos[2] + refCam->C[i]; /* :944 */
    double H[PO_MAX_VIS * 9];
    po_homographies(s, patch, center, normal, H); /* :948 */

    /* the first occurrence of the reference camera is served by the window; the others are tapped in camIdx order */
    int firstRef = -1, other[PO_MAX_VIS], M = 0;
    for (int i = 0; i < K; ++i) {
        if (firstRef < 0 && patch->camIdx[i] == patch->refCamIdx) firstRef = i;
        else other[M++] = i;
    }
    const int hasRef = firstRef >= 0;
    const double invK = 1.0 / (double)K, invDiffW = 1.0 / s->cfg.diffWeighting;
    const double a0 = pt[0] - r, b0 = pt[1] - r;
    /* corners_inside (pais_eval.hpp): if every other camera maps the four window corners into [3, w-4) x [3, h-4) with a
     * denominator of one sign, no tap of the (convex) window can leave the image: the per-tap test of :999 is skipped
     * for this evaluation; otherwise it is applied tap by tap as the reference does */
    int fast = 1;
    for (int i = 0; i < M && fast; ++i) {
        const double *Hi = H + 9 * other[i];
        const po_camera *cam = &s->cams[patch->camIdx[other[i]]];
        int npos = 0, nneg = 0;
        for (int corner = 0; corner < 4; ++corner) {
            const double x = a0 + (double)((corner & 1) ? (S - 1) : 0), y = b0 + (double)((corner & 2) ? (S - 1) : 0);
            const double w = fma(Hi[7], y, fma(Hi[6], x, Hi[8]));
            const double rw = 1.0 / w;
            const double ix = fma(Hi[1], y, fma(Hi[0], x, Hi[2])) * rw, iy = fma(Hi[4], y, fma(Hi[3], x, Hi[5])) * rw;
            /* (int) of a NaN / out-of-range double is 0 / saturated on the GPU: rejected either way */
            /* (the box shrunk by one pixel and a denominator of ordinary size: pais_eval.hpp corners_inside) */
            const int okx = ix >= 3 && ix < (double)(cam->width[LOD] - 4), oky = iy >= 3 && iy < (double)(cam->height[LOD] - 4);
            if (!(okx && oky) || !(fabs(w) > 1e-90 && fabs(w) < 1e90)) fast = 0;
            npos += w > 0.0;
            nneg += w < 0.0;
        }
        if (!(npos == 4 || nneg == 4)) fast = 0;
    }
    double pf[4][64] = {{0}}, pw[4][64] = {{0}};
    double c[PO_MAX_VIS];
    for (int k = 0; k < S * S; ++k) {
        const int yi = k / S, xi = k - yi * S;
        const double x = a0 + (double)xi, y = b0 + (double)yi;
        const int rx = cv_round(x), ry = cv_round(y);
        if (refImg[(size_t)ry * refCols + rx] == 0) continue; /* :986 */
        const int qx0 = (int)x, qy0 = (int)y;
        const double refCol = lerp3_u8(refImg, refCols, qx0, qy0, x - (double)qx0, y - (double)qy0);
        double ws = s->cfg.adaptiveDistanceEnable ? s->gauss[xi * S + yi] : 1.0;
        if (s->cfg.adaptiveGradientEnable) ws *= po_det_exp_poly(-1.0 / (edgeImg[(size_t)ry * refCols + rx] * s->cfg.gradientWeighting));
        /* round 6: patches seen by 13 or more cameras sum their colours (and the absolute deviations below) in TWO groups --
         * the reference colour and the first h = 2 * ((M + 4) / 4) other cameras, then the remaining ones, each group sequentially,
         * the two group sums added once -- so that two waves of the HIP path can each own a group (pais_tile2.hpp) */
        const int twoLevel = K >= PO_TWO_LEVEL_K;
        const int hSplit = twoLevel ? 2 * ((M + 4) / 4) : M;
        double sum = hasRef ? refCol : 0.0, sum2 = 0.0;
        for (int i0 = 0; i0 < M;) {
            const int left = M - i0;
            const int g = (left >= 4 || left == 2) ? 2 : (left == 3 ? 3 : 1);
            double ww[3], nx[3], ny[3], rw[3];
            for (int u = 0; u < g; ++u) {
                const double *Hi = H + 9 * other[i0 + u];
                ww[u] = fma(Hi[7], y, fma(Hi[6], x, Hi[8]));
                nx[u] = fma(Hi[1], y, fma(Hi[0], x, Hi[2]));
                ny[u] = fma(Hi[4], y, fma(Hi[3], x, Hi[5]));
            }
            if (g == 3) {
                const double p01 = ww[0] * ww[1];
                const double rr = 1.0 / (p01 * ww[2]);
                rw[2] = rr * p01;
                const double r01 = rr * ww[2];
                rw[0] = r01 * ww[1];
                rw[1] = r01 * ww[0];
            } else if (g == 2) {
                const double rr = 1.0 / (ww[0] * ww[1]);
                rw[0] = rr * ww[1];
                rw[1] = rr * ww[0];
            } else {
                rw[0] = 1.0 / ww[0];
            }
            for (int u = 0; u < g; ++u) {
                const po_camera *cam = &s->cams[patch->camIdx[other[i0 + u]]];
                const int cols = cam->width[LOD], rows = cam->height[LOD];
                const double jx = nx[u] * rw[u], jy = ny[u] * rw[u];
                if (!fast && !(jx >= 2 && jx < cols - 3 && jy >= 2 && jy < rows - 3)) return DBL_MAX; /* :999 -- whole call */
                const int qx = (int)jx, qy = (int)jy;
                c[i0 + u] = lerp3_u8(cam->img[LOD], cols, qx, qy, jx - (double)qx, jy - (double)qy);
                if (i0 + u < hSplit) sum += c[i0 + u];
                else sum2 += c[i0 + u];
            }
            i0 += g;
        }
        if (twoLevel) sum += sum2;
        const double mean = sum * invK;
        double sad = hasRef ? fabs(refCol - mean) : 0.0, sad2 = 0.0;
        for (int i = 0; i < hSplit; ++i) sad += fabs(c[i] - mean);
        for (int i = hSplit; i < M; ++i) sad2 += fabs(c[i] - mean);
        if (twoLevel) sad += sad2;
        sad *= invK;
        double weight = ws;
        if (s->cfg.adaptiveDifferenceEnable) weight *= po_det_exp_poly(-(sad * sad) * invDiffW);
        const int a = (k >> 6) & 3, l = k & 63;
        pw[a][l] += weight;
        pf[a][l] = fma(weight, sad, pf[a][l]);
    }
    const double F = ((tree64(pf[0]) + tree64(pf[1])) + tree64(pf[2])) + tree64(pf[3]);
    const double W = ((tree64(pw[0]) + tree64(pw[1])) + tree64(pw[2])) + tree64(pw[3]);
    return F / W;
}

/* ------------------------------------------------------------------------ */
/* cost: PAIS::getFitness, patch.cpp:914-1047                                */
/* ------------------------------------------------------------------------ */
double po_get_fitness(const po_scene *s, const po_patch *patch, const double pos[3])
{
    if ((s->detMath || s->treeSum) && !s->costLiteral) return get_fitness_kernel(s, patch, pos); /* the two switches are set together */
    const int patchRadius = s->cfg.patchRadius;
    const int LOD = patch->LOD;
    const po_camera *refCam = &s->cams[patch->refCamIdx];
    const int camNum = patch->numCam;
    const double *edgeImg = refCam->edge[LOD];
    const uint8_t *refImg = refCam->img[LOD];
    const int refCols = refCam->width[LOD], refRows = refCam->height[LOD];

    double normal[3];
    double sph[2] = {pos[0], pos[1]};
    s2n(s, sph, normal);

    if (dot3(normal, refCam->optN) > 0) return DBL_MAX; /* :939 */

    double center[3];
    for (int i = 0; i < 3; ++i) center[i] = patch->ray[i] * pos[2] + refCam->C[i]; /* :944 */

    double H[PO_MAX_VIS * 9];
    po_homographies(s, patch, center, normal, H); /* :948 */

    double pt[2];
    if (!po_project(s, patch->refCamIdx, center, pt, LOD)) return DBL_MAX; /* :952 */

    if (pt[0] - patchRadius < 2 || pt[0] + patchRadius >= refCols - 3 ||
        pt[1] - patchRadius < 2 || pt[1] + patchRadius >= refRows - 3)
        return DBL_MAX; /* :957-962 (edgeImg dims == level dims) */

    fit_px_ctx fx;
    fx.s = s; fx.patch = patch; fx.H = H; fx.refImg = refImg; fx.edgeImg = edgeImg; fx.refCols = refCols;
    fx.LOD = LOD; fx.camNum = camNum;
    double fitness = 0;
    double sumWeight = 0;
    double weight, avgSad;

    /* the reference's walk: x outer, y inner, sequential sums (patch.cpp:979-1041) */
    const double *it = s->gauss;
    if (s->literalVariant & 2) {
        /* CONTROL (not the reference): the same pixels with the same coordinates and weights, accumulated y outer / x inner.
         * x and y of a pixel are the values the reference's two loops give it (start + k by repeated ++). */
        const int S = s->cfg.patchSize;
        double xs[PO_MAX_PATCH_SIZE], ys[PO_MAX_PATCH_SIZE];
        int nx = 0, ny = 0;
        for (double x = pt[0] - patchRadius; x <= pt[0] + patchRadius && nx < PO_MAX_PATCH_SIZE; ++x) xs[nx++] = x;
        for (double y = pt[1] - patchRadius; y <= pt[1] + patchRadius && ny < PO_MAX_PATCH_SIZE; ++y) ys[ny++] = y;
        /* an overflowing tap makes the whole call DBL_MAX in either order */
        for (int yi = 0; yi < ny; ++yi)
            for (int xi = 0; xi < nx; ++xi) {
                int st = fit_pixel(&fx, xs[xi], ys[yi], it[xi * S + yi], &weight, &avgSad);
                if (st < 0) return DBL_MAX;
                if (st == 0) continue;
                sumWeight += weight;
                if (s->literalVariant & 4) fitness = fma(weight, avgSad, fitness);
                else fitness += weight * avgSad;
            }
        if (s->literalVariant & 1) return fitness * (1.0 / sumWeight);
        return fitness / sumWeight;
    }
    for (double x = pt[0] - patchRadius; x <= pt[0] + patchRadius; ++x) {
        for (double y = pt[1] - patchRadius; y <= pt[1] + patchRadius; ++y, ++it) {
            int st = fit_pixel(&fx, x, y, *it, &weight, &avgSad);
            if (st < 0) return DBL_MAX;
            if (st == 0) continue;
            sumWeight += weight;
            if (s->literalVariant & 4) fitness = fma(weight, avgSad, fitness); /* control: contracted */
            else fitness += weight * avgSad;
        }
    }
    if (s->literalVariant & 1) return fitness * (1.0 / sumWeight); /* control: ONE rounding of the call perturbed */
    return fitness / sumWeight;
}

/* ------------------------------------------------------------------------ */
/* PSO: pso/psosolver.cpp + pso/particle.cpp                                 */
/* ------------------------------------------------------------------------ */
#define PO_PSO_MAXN 256
#define PO_PSO_MAXD 8
typedef struct {
    double pBest[PO_PSO_MAXD], nBest[PO_PSO_MAXD], pos[PO_PSO_MAXD], vec[PO_PSO_MAXD];
    const double *lBest;
    double fitness, pBestFitness;
} po_particle;

typedef struct { double dist; int idx; } po_local;

void po_pso_run(int dim, const double *rangeL_, const double *rangeU_,
                po_fitness_fn fn, void *obj, int maxIteration, int particleNum,
                const double *init, po_rand_fn rnd, void *rngObj, int omp,
                po_pso_result *res, double *trace, int traceCap, int *traceLen)
{
    /* psosolver.h:104-111 defaults */
    const double convergenceThreshold = 0.01;
    double iw = 0.8;
    const double pw = 1.2, gw = 1.5, lw = 1.0, nw = 1.0;
    const double minIw = 0.4;
    int localK = 5;
    if (particleNum > PO_PSO_MAXN) particleNum = PO_PSO_MAXN;
    if (dim > PO_PSO_MAXD) dim = PO_PSO_MAXD;
    localK = particleNum < localK ? particleNum : localK; /* :26 */

    double rangeL[PO_PSO_MAXD], rangeU[PO_PSO_MAXD], rangeInter[PO_PSO_MAXD];
    for (int i = 0; i < dim; i++) { /* :35-39 */
        rangeL[i] = rangeL_[i];
        rangeU[i] = rangeU_[i];
        rangeInter[i] = rangeU_[i] - rangeL_[i];
    }
    po_particle *P = (po_particle *)calloc((size_t)particleNum, sizeof(po_particle));
    for (int i = 0; i < particleNum; ++i) { /* particle.cpp:5-20 */
        P[i].fitness = 1.7976931348623158e+308;
        P[i].pBestFitness = 1.7976931348623158e+308;
        P[i].lBest = NULL;
    }
#define RANDOM() (((double)rnd(rngObj)) / ((double)2147483647)) /* :66-68, glibc RAND_MAX */

    /* initParticles :94-110 */
    for (int d = 0; d < dim; d++) {
        for (int i = 0; i < particleNum; i++) {
            P[i].pos[d] = (rangeInter[d] * RANDOM()) + rangeL[d];
            P[i].vec[d] = (2.0 * rangeInter[d] * RANDOM()) - rangeInter[d];
            P[i].pBest[d] = P[i].pos[d];
        }
    }
    /* setParticle(init) :267-284, vec == NULL, idx == 0 */
    if (init) {
        for (int d = 0; d < dim; d++) {
            P[0].pos[d] = init[d];
            P[0].pBest[d] = P[0].pos[d];
            P[0].vec[d] = (2.0 * rangeInter[d] * RANDOM()) - rangeInter[d];
        }
    }

    int tl = 0;
    int evals = 0;
    /* run() :286-306 ; initFitness :112-119 */
#pragma omp parallel for if (omp) schedule(static)
    for (int i = 0; i < particleNum; i++) {
        P[i].fitness = fn(P[i].pos, obj);
        P[i].pBestFitness = P[i].fitness;
    }
    evals += particleNum;
    int gIdx = 0; /* gBest = particles[0].pBest (a POINTER into the particle, psosolver.h:54) */
    double gBestFitness = P[0].pBestFitness;
#define UPDATE_GBEST()                                                     \
    for (int j = 0; j < particleNum; j++) {                                \
        if (P[j].pBestFitness <= gBestFitness) { /* :142, last min wins */ \
            gBestFitness = P[j].pBestFitness;                              \
            gIdx = j;                                                      \
        }                                                                  \
    }                                                                      \
    gSig = (gSig ^ (uint64_t)(gIdx + 1)) * 1099511628211ULL; /* which particle owns gBest, iteration by iteration */
    /* (diagnosis only, not part of the reference) gSig folds EVERY discrete decision of the run: the owner of gBest after each
     * updateGbest, which particles improved their pBest in each updateFitness (:128), every particle's lBest and per-dimension
     * nBest selection in moveParticles (:151-218), the iteration count.  Two runs with equal signatures differ by rounding only. */
    uint64_t gSig = 1469598103934665603ULL;
    unsigned char improved[PO_PSO_MAXN];
    UPDATE_GBEST();

    int iteration;
    for (iteration = 0; iteration < maxIteration; iteration++) {
        const double *gBest = P[gIdx].pBest;
        /* getDispersionIDX :70-80, getVelocityIDX :82-92 */
        double disp = 0, vel = 0;
        for (int i = 0; i < particleNum; i++)
            for (int j = 0; j < dim; j++) disp += fabs(P[i].pos[j] - gBest[j]);
        disp /= (dim * particleNum);
        if (disp < convergenceThreshold) {
            for (int i = 0; i < particleNum; i++)
                for (int j = 0; j < dim; j++) vel += fabs(P[i].vec[j]);
            vel /= (dim * particleNum);
            if (vel < convergenceThreshold) break; /* :295 */
        }

        /* moveParticles :220-265 (serial draw order, 1 thread) */
        for (int i = 0; i < particleNum; i++) {
            double pVecW, gVecW, lVecW, nVecW;
            po_particle *p = &P[i];
            pVecW = pw * RANDOM();
            gVecW = gw * RANDOM();
            lVecW = lw * RANDOM();
            nVecW = nw * RANDOM();

            /* getLocalBest(i) :151-191 */
            {
                po_local cont[PO_PSO_MAXN];
                const double *pp = P[i].pBest;
                for (int k = 0; k < particleNum; k++) {
                    cont[k].dist = 0;
                    cont[k].idx = k;
                    if (k == i) { cont[k].dist = DBL_MAX; continue; }
                    for (int d = 0; d < dim; d++)
                        cont[k].dist += (pp[d] - P[k].pBest[d]) * (pp[d] - P[k].pBest[d]);
                }
                /* std::sort(dist <): MSVC (n<=32) and libstdc++ (n<=16) run a
                 * plain insertion sort == stable; restated as stable. */
                for (int a = 1; a < particleNum; ++a) {
                    po_local v = cont[a];
                    int b = a - 1;
                    while (b >= 0 && v.dist < cont[b].dist) { cont[b + 1] = cont[b]; --b; }
                    cont[b + 1] = v;
                }
                double minFitness = DBL_MAX;
                const double *lBest = pp;
                int lSel = i;
                for (int k = 0; k < localK; k++) {
                    const po_particle *q = &P[cont[k].idx];
                    if (q->pBestFitness < minFitness) {
                        minFitness = q->pBestFitness;
                        lBest = q->pBest;
                        lSel = cont[k].idx;
                    }
                }
                p->lBest = lBest;
                gSig = (gSig ^ (uint64_t)(lSel + 1)) * 1099511628211ULL;
            }
            /* setNearNeighborBest(i) :193-218 */
            {
                const double fitness = P[i].fitness;
                const double *ppos = P[i].pos;
                double *nBest = P[i].nBest;
                for (int d = 0; d < dim; d++) {
                    double maxFDR = -DBL_MAX;
                    int nSel = -1;
                    for (int k = 0; k < particleNum; k++) {
                        if (k == i) continue;
                        double FDR = (fitness - P[k].pBestFitness) / fabs(ppos[d] - P[k].pBest[d]);
                        if (FDR > maxFDR) {
                            maxFDR = FDR;
                            nBest[d] = P[k].pBest[d];
                            nSel = k;
                        }
                    }
                    gSig = (gSig ^ (uint64_t)(nSel + 2)) * 1099511628211ULL;
                }
            }
            for (int d = 0; d < dim; d++) {
                p->vec[d] = iw * p->vec[d] +
                            pVecW * (p->pBest[d] - p->pos[d]) +
                            gVecW * (gBest[d] - p->pos[d]) +
                            lVecW * (p->lBest[d] - p->pos[d]) +
                            nVecW * (p->nBest[d] - p->pos[d]);
                p->pos[d] += p->vec[d];
                if (p->pos[d] > rangeU[d]) p->pos[d] = rangeU[d];
                if (p->pos[d] < rangeL[d]) p->pos[d] = rangeL[d];
            }
        }

        /* updateFitness :121-135 */
#pragma omp parallel for if (omp) schedule(static)
        for (int i = 0; i < particleNum; i++) {
            po_particle *p = &P[i];
            p->fitness = fn(p->pos, obj);
            improved[i] = 0;
            if (p->fitness < p->pBestFitness) {
                p->pBestFitness = p->fitness;
                for (int d = 0; d < dim; d++) p->pBest[d] = p->pos[d];
                improved[i] = 1;
            }
        }
        evals += particleNum;
        for (int i = 0; i < particleNum; i++) gSig = (gSig ^ (uint64_t)(improved[i] + 1)) * 1099511628211ULL;
        UPDATE_GBEST();

        iw = (iw - 1.0 / maxIteration) > minIw ? (iw - 1.0 / maxIteration) : minIw; /* :304 */

        if (trace) {
            for (int i = 0; i < particleNum && tl + 11 <= traceCap; ++i) {
                for (int d = 0; d < 3; ++d) trace[tl++] = d < dim ? P[i].pos[d] : 0;
                for (int d = 0; d < 3; ++d) trace[tl++] = d < dim ? P[i].vec[d] : 0;
                for (int d = 0; d < 3; ++d) trace[tl++] = d < dim ? P[i].pBest[d] : 0;
                trace[tl++] = P[i].fitness;
                trace[tl++] = P[i].pBestFitness;
            }
            if (tl + 2 <= traceCap) { trace[tl++] = gIdx; trace[tl++] = iw; }
        }
    }
    if (traceLen) *traceLen = tl;
    for (int d = 0; d < dim && d < 3; ++d) res->gBest[d] = P[gIdx].pBest[d];
    res->gBestFitness = gBestFitness;
    res->iterations = iteration;
    res->evals = evals;
    res->gbestSig = (gSig ^ (uint64_t)(iteration + 1)) * 1099511628211ULL;
    free(P);
#undef RANDOM
#undef UPDATE_GBEST
}

/* ------------------------------------------------------------------------ */
/* patch state machine: mvs/patch.cpp, mvs/abstractpatch.cpp                 */
/* ------------------------------------------------------------------------ */
/* abstractpatch.cpp:24-41 */
static void patch_init(po_patch *p)
{
    memset(p, 0, sizeof(*p));
    p->id = -1;
    p->refCamIdx = -1;
    p->LOD = -1;
    p->fitness = DBL_MAX;
    p->priority = DBL_MAX;
    p->correlation = 0;
    p->expanded = 0;
}
/* abstractpatch.cpp:43-46 */
static void set_normal3(po_patch *p, const double n[3])
{
    for (int i = 0; i < 3; ++i) p->normal[i] = n[i];
    po_normal2spherical(p->normal, p->normalS);
}
/* abstractpatch.cpp:48-51 */
static void set_normal2(const po_scene *s, po_patch *p, const double ns[2])
{
    p->normalS[0] = ns[0];
    p->normalS[1] = ns[1];
    s2n(s, p->normalS, p->normal);
}

/* patch.cpp:6-23 */
int po_is_neighbor(const po_scene *s, const po_patch *a, const po_patch *b)
{
    double dlt[3];
    for (int i = 0; i < 3; ++i) dlt[i] = a->center[i] - b->center[i];
    double dist = 0;
    dist += fabs(dot3(dlt, a->normal));
    dist += fabs(dot3(dlt, b->normal));
    return dist <= s->cfg.neighborRadius;
}

/* patch.cpp:390-413 */
void po_set_estimated_normal(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    const int camNum = p->numCam;
    if (camNum < s->cfg.minCamNum) { p->drop = 1; return; }
    double dir[3], normal[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < camNum; i++) {
        const po_camera *cam = &s->cams[p->camIdx[i]];
        for (int k = 0; k < 3; ++k) dir[k] = cam->C[k] - p->center[k];
        double sc = (1.0 / norm3(dir));
        for (int k = 0; k < 3; ++k) dir[k] *= sc;
        for (int k = 0; k < 3; ++k) normal[k] += dir[k];
    }
    double sc = (1.0 / norm3(normal));
    for (int k = 0; k < 3; ++k) normal[k] *= sc;
    set_normal3(p, normal);
}

/* patch.cpp:26-34 */
void po_patch_init_seed(const po_scene *s, po_patch *p, const double center[3],
                        int numCam, const int *camIdx, uint64_t key)
{
    patch_init(p);
    p->type = PO_TYPE_SEED;
    for (int i = 0; i < 3; ++i) p->center[i] = center[i];
    p->numCam = numCam > PO_MAX_VIS ? PO_MAX_VIS : numCam;
    for (int i = 0; i < p->numCam; ++i) p->camIdx[i] = camIdx[i];
    p->drop = 0;
    p->key = key;
    po_set_estimated_normal(s, p);
}

/* patch.cpp:723-761 */
void po_expand_visible_camera(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    int exp[PO_MAX_VIS * 2];
    int n = 0;
    for (int i = 0; i < s->numCams; ++i) {
        const po_camera *cam = &s->cams[i];
        double neg[3] = {-cam->optN[0], -cam->optN[1], -cam->optN[2]};
        if (dot3(p->normal, neg) >= s->cfg.visibleCorrelation) {
            if (n < PO_MAX_VIS) exp[n++] = i;
        }
    }
    if (n < s->cfg.minCamNum) {
        for (int i = 0; i < p->numCam; ++i) {
            const po_camera *cam = &s->cams[p->camIdx[i]];
            double neg[3] = {-cam->optN[0], -cam->optN[1], -cam->optN[2]};
            if (dot3(p->normal, neg) >= s->cfg.visibleCorrelation / 2.0) {
                if (n < PO_MAX_VIS * 2) exp[n++] = p->camIdx[i];
            }
        }
        /* sort + unique */
        for (int a = 1; a < n; ++a) {
            int v = exp[a], b = a - 1;
            while (b >= 0 && exp[b] > v) { exp[b + 1] = exp[b]; --b; }
            exp[b + 1] = v;
        }
        int m = 0;
        for (int a = 0; a < n; ++a)
            if (m == 0 || exp[m - 1] != exp[a]) exp[m++] = exp[a];
        n = m;
    }
    if (n > PO_MAX_VIS) n = PO_MAX_VIS;
    p->numCam = n;
    for (int i = 0; i < n; ++i) p->camIdx[i] = exp[i];
    if (p->numCam < s->cfg.minCamNum) p->drop = 1;
}

/* patch.cpp:36-43 */
void po_patch_init_expand(const po_scene *s, po_patch *p, const double center[3],
                          const double parentNormal[3], int parentNumCam,
                          const int *parentCamIdx, uint64_t key)
{
    patch_init(p);
    p->type = PO_TYPE_EXPAND;
    for (int i = 0; i < 3; ++i) p->center[i] = center[i];
    p->numCam = parentNumCam > PO_MAX_VIS ? PO_MAX_VIS : parentNumCam;
    for (int i = 0; i < p->numCam; ++i) p->camIdx[i] = parentCamIdx[i];
    p->drop = 0;
    p->key = key;
    set_normal3(p, parentNormal);
    po_expand_visible_camera(s, p);
}

/* patch.cpp:415-445 */
void po_set_reference_camera(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    const int camNum = p->numCam;
    if (camNum < s->cfg.minCamNum) { p->drop = 1; return; }
    p->refCamIdx = -1;
    double maxCorr = -DBL_MAX;
    double corr;
    for (int i = 0; i < camNum; i++) {
        const po_camera *cam = &s->cams[p->camIdx[i]];
        double neg[3] = {-cam->optN[0], -cam->optN[1], -cam->optN[2]};
        corr = dot3(p->normal, neg);
        if (corr > maxCorr) {
            maxCorr = corr;
            p->refCamIdx = p->camIdx[i];
        }
    }
    if (p->refCamIdx < 0) {
        p->refCamIdx = p->camIdx[0];
        p->drop = 1;
    }
}

/* patch.cpp:447-461 */
void po_set_depth_and_ray(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    if (p->refCamIdx < 0) { p->drop = 1; return; }
    const po_camera *rc = &s->cams[p->refCamIdx];
    for (int i = 0; i < 3; ++i) p->ray[i] = p->center[i] - rc->C[i];
    p->depth = norm3(p->ray);
    double sc = (1.0 / p->depth);
    for (int i = 0; i < 3; ++i) p->ray[i] = p->ray[i] * sc;
}

/* patch.cpp:463-509 */
void po_set_depth_range(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    const int camNum = p->numCam;
    if (camNum < s->cfg.minCamNum) { p->drop = 1; return; }
    const po_camera *refCam = &s->cams[p->refCamIdx];
    double c2[3];
    for (int i = 0; i < 3; ++i) c2[i] = p->ray[i] * (p->depth + 1.0) + refCam->C[i];
    double p1[2], p2[2];
    double worldDist, imgDist;
    double maxWorldDist = -DBL_MAX;
    for (int i = 0; i < camNum; i++) {
        if (p->camIdx[i] == p->refCamIdx) continue;
        po_project(s, p->camIdx[i], p->center, p1, 0);
        po_project(s, p->camIdx[i], c2, p2, 0);
        double dx = p1[0] - p2[0], dy = p1[1] - p2[1];
        imgDist = sqrt(dx * dx + dy * dy);
        worldDist = 1.0 / imgDist;
        if (worldDist > maxWorldDist && imgDist >= 0.01) maxWorldDist = worldDist;
    }
    if (maxWorldDist == -DBL_MAX) { p->drop = 1; return; }
    double a = p->depth - maxWorldDist * s->cfg.depthRangeScalar;
    p->depthRange[0] = a > 0.0 ? a : 0.0; /* max(a, 0.0) */
    double b = maxWorldDist * s->cfg.depthRangeScalar, cc = s->cfg.neighborRadius * 100;
    p->depthRange[1] = p->depth + (cc < b ? cc : b); /* min(b, cc) */
}

/* patch.cpp:511-610 */
void po_set_lod(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    if (p->refCamIdx < 0) { p->drop = 1; return; }
    const int patchRadius = s->cfg.patchRadius;
    const po_camera *refCam = &s->cams[p->refCamIdx];
    double mean = 0, variance = 0;
    int count;
    double pt[2];
    const int size = s->cfg.patchSize;
    uint8_t *textures = (uint8_t *)malloc((size_t)size * size);

    p->LOD = s->cfg.minLOD - 1;
    while (variance < s->cfg.textureVariation) {
        p->LOD++;
        if (p->LOD >= refCam->maxLOD) {
            p->LOD = refCam->maxLOD;
            free(textures);
            return;
        }
        if (!po_project(s, p->refCamIdx, p->center, pt, p->LOD)) {
            p->LOD = (p->LOD - 1) > 0 ? (p->LOD - 1) : 0;
            free(textures);
            return;
        }
        mean = 0; variance = 0; count = 0;
        const uint8_t *img = refCam->img[p->LOD];
        const int cols = refCam->width[p->LOD];
        for (int x = cv_round(pt[0]) - patchRadius; x <= cv_round(pt[0]) + patchRadius; x++) {
            for (int y = cv_round(pt[1]) - patchRadius; y <= cv_round(pt[1]) + patchRadius; y++) {
                if (!in_image_i(refCam, x, y, p->LOD)) {
                    p->LOD = (p->LOD - 1) > 0 ? (p->LOD - 1) : 0;
                    free(textures);
                    return;
                }
                textures[count] = img[(size_t)y * cols + x];
                mean += textures[count];
                count++;
            }
        }
        mean /= count;
        if (!s->treeSum) {
            for (int i = 0; i < count; i++) variance += (textures[i] - mean) * (textures[i] - mean);
        } else {
            /* kernel order: window pixel k = yi*S + xi on lane k%64, then the butterfly */
            double pv[64] = {0};
            const int cx = cv_round(pt[0]), cy = cv_round(pt[1]);
            for (int k = 0; k < size * size; ++k) {
                int yi = k / size, xi = k - yi * size;
                double t = (double)img[(size_t)(cy - patchRadius + yi) * cols + (cx - patchRadius + xi)];
                pv[k & 63] += (t - mean) * (t - mean);
            }
            variance = tree64(pv);
        }
        variance /= count;
    }
    free(textures);
}

/* patch.cpp:612-625 */
void po_set_priority(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    double w1 = 1.0, w2 = 1.0;
    const int totalCamNum = s->numCams;
    const int camNum = p->numCam;
    double camRatio = ((double)camNum) / ((double)totalCamNum);
    p->priority = p->fitness * m_exp(s, -p->correlation / w1 - camRatio / w2) * (p->LOD + 1.0);
}

/* patch.cpp:627-653 (colour pick from the RGB image is host-side, not restated) */
void po_set_image_point(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    const int camNum = p->numCam;
    if (camNum == 0) return;
    for (int i = 0; i < camNum; ++i) po_project(s, p->camIdx[i], p->center, p->imgPoint[i], 0);
}

typedef po_fit_ctx fit_ctx;
double po_fit_cb(const double *pos, void *obj)
{
    fit_ctx *c = (fit_ctx *)obj;
    return po_get_fitness(c->s, c->p, pos);
}
#define fit_cb po_fit_cb
typedef po_rng_ctx rng_ctx;
uint32_t po_rng_cb(void *o)
{
    rng_ctx *r = (rng_ctx *)o;
    return po_rand31(r->seed, r->key, r->run, r->k++);
}
#define rng_cb po_rng_cb

/* patch.cpp:180-219 */
void po_pso_optimization(const po_scene *s, po_patch *p)
{
    double rangeL[] = {0.0, p->normalS[1] - M_PI / 2.0, p->depthRange[0]};
    double rangeU[] = {M_PI, p->normalS[1] + M_PI / 2.0, p->depthRange[1]};
    double init[] = {p->normalS[0], p->normalS[1], p->depth};
    int maxIt, pn;
    if (p->type == PO_TYPE_SEED) {
        maxIt = s->cfg.maxIteration * 2;
        pn = s->cfg.particleNum * 2;
    } else {
        double lo = p->normalS[0] - M_PI / s->cfg.reduceNormalRange;
        double hi = p->normalS[0] + M_PI / s->cfg.reduceNormalRange;
        rangeL[0] = 0.0 < lo ? lo : 0.0;   /* max(0.0, lo) */
        rangeU[0] = hi < M_PI ? hi : M_PI; /* min(M_PI, hi) */
        rangeL[1] = p->normalS[1] - M_PI / s->cfg.reduceNormalRange;
        rangeU[1] = p->normalS[1] + M_PI / s->cfg.reduceNormalRange;
        maxIt = s->cfg.maxIteration;
        pn = s->cfg.particleNum;
    }
    fit_ctx fc = {s, p};
    rng_ctx rc = {s->seed, p->key, (uint32_t)p->psoRuns, 0};
    po_pso_result res;
    po_pso_run(3, rangeL, rangeU, fit_cb, &fc, maxIt, pn, init, rng_cb, &rc, s->ompParticles, &res, NULL, 0, NULL);
    p->psoRuns++;
    p->psoIters += res.iterations;
    p->psoEvals += res.evals;
    p->psoSig = (p->psoSig ^ res.gbestSig) * 1099511628211ULL;

    p->fitness = res.gBestFitness;
    double ns[2] = {res.gBest[0], res.gBest[1]};
    set_normal2(s, p, ns);
    p->depth = res.gBest[2];
    const po_camera *rc2 = &s->cams[p->refCamIdx];
    for (int i = 0; i < 3; ++i) p->center[i] = p->ray[i] * p->depth + rc2->C[i];
}

/* patch.cpp:332-386 ; returns 0 when the patch was dropped */
static int homography_patch(const po_scene *s, po_patch *p, const double pt[2], int camI,
                            const double *H, double *hp)
{
    if (p->drop) return 0;
    const int patchRadius = s->cfg.patchRadius;
    const po_camera *cam = &s->cams[camI];
    const uint8_t *img = cam->img[p->LOD];
    const int cols = cam->width[p->LOD], rows = cam->height[p->LOD];
    double w, ix, iy;
    int px[4], py[4];
    int count = 0;
    double sum = 0;
    if (s->treeSum) {
        /* kernel order: hp index k = yi*S + xi, squared norm by lane partials + butterfly */
        const int S = s->cfg.patchSize;
        const double a0 = pt[0] - patchRadius, b0 = pt[1] - patchRadius;
        double ps[64] = {0};
        for (int k = 0; k < S * S; ++k) {
            const int yi = k / S, xi = k - yi * S;
            const double x = a0 + (double)xi, y = b0 + (double)yi;
            w = (H[6] * x + H[7] * y + H[8]);
            ix = (H[0] * x + H[1] * y + H[2]) / w;
            iy = (H[3] * x + H[4] * y + H[5]) / w;
            if (ix < 0 || ix >= cols - 1 || iy < 0 || iy >= rows - 1 || w == 0 || isnan(ix) || isnan(iy)) {
                p->drop = 1;
                return 0;
            }
            px[0] = (int)ix; py[0] = (int)iy;
            px[1] = px[0] + 1; py[2] = py[0] + 1;
            double v = (double)img[(size_t)py[0] * cols + px[0]] * (px[1] - ix) * (py[2] - iy) +
                       (double)img[(size_t)py[0] * cols + px[1]] * (ix - px[0]) * (py[2] - iy) +
                       (double)img[(size_t)py[2] * cols + px[0]] * (px[1] - ix) * (iy - py[0]) +
                       (double)img[(size_t)py[2] * cols + px[1]] * (ix - px[0]) * (iy - py[0]);
            hp[k] = v;
            ps[k & 63] += v * v;
        }
        double inv = 1.0 / sqrt(tree64(ps));
        for (int k = 0; k < S * S; ++k) hp[k] = hp[k] * inv;
        return 1;
    }
    for (double x = pt[0] - patchRadius; x <= pt[0] + patchRadius; ++x) {
        for (double y = pt[1] - patchRadius; y <= pt[1] + patchRadius; ++y) {
            w = (H[6] * x + H[7] * y + H[8]);
            ix = (H[0] * x + H[1] * y + H[2]) / w;
            iy = (H[3] * x + H[4] * y + H[5]) / w;
            if (ix < 0 || ix >= cols - 1 || iy < 0 || iy >= rows - 1 || w == 0 || p->drop ||
                isnan(ix) || isnan(iy)) {
                p->drop = 1;
                return 0;
            }
            px[0] = (int)ix; py[0] = (int)iy;
            px[1] = px[0] + 1; py[1] = py[0];
            px[2] = px[0]; py[2] = py[0] + 1;
            px[3] = px[0] + 1; py[3] = py[0] + 1;
            hp[count] = (double)img[(size_t)py[0] * cols + px[0]] * (px[1] - ix) * (py[2] - iy) +
                        (double)img[(size_t)py[1] * cols + px[1]] * (ix - px[0]) * (py[2] - iy) +
                        (double)img[(size_t)py[2] * cols + px[2]] * (px[1] - ix) * (iy - py[0]) +
                        (double)img[(size_t)py[3] * cols + px[3]] * (ix - px[0]) * (iy - py[0]);
            sum += hp[count] * hp[count];
            ++count;
        }
    }
    /* hp /= sqrt(sum): OpenCV Mat /= double scales by 1./s */
    double inv = 1.0 / sqrt(sum);
    for (int i = 0; i < count; ++i) hp[i] = hp[i] * inv;
    return 1;
}

/* patch.cpp:221-267 */
void po_set_correlation_table(const po_scene *s, po_patch *p, const double *H)
{
    const int camNum = p->numCam;
    const int S2 = s->cfg.patchSize * s->cfg.patchSize;
    for (int i = 0; i < camNum * camNum; ++i) p->corrTable[i] = 0;

    double pt[2];
    po_project(s, p->refCamIdx, p->center, pt, p->LOD);

    double *HP = (double *)malloc(sizeof(double) * (size_t)S2 * (size_t)(camNum > 0 ? camNum : 1));
    for (int i = 0; i < camNum; i++) homography_patch(s, p, pt, p->camIdx[i], H + 9 * i, HP + (size_t)S2 * i);

    if (p->drop) {
        p->correlation = 0;
        free(HP);
        return;
    }
    for (int i = 0; i < camNum; ++i) {
        p->corrTable[i * camNum + i] = 0;
        for (int j = i + 1; j < camNum; ++j) {
            double corr = 0;
            const double *a = HP + (size_t)S2 * i, *b = HP + (size_t)S2 * j;
            if (!s->treeSum) {
                for (int k = 0; k < S2; ++k) corr += a[k] * b[k];
            } else {
                double pc[64] = {0};
                for (int k = 0; k < S2; ++k) pc[k & 63] += a[k] * b[k];
                corr = tree64(pc);
            }
            p->corrTable[i * camNum + j] = corr;
            p->corrTable[j * camNum + i] = corr;
        }
    }
    p->correlation = 0;
    for (int i = 0; i < camNum; ++i)
        for (int j = 0; j < camNum; ++j) p->correlation += p->corrTable[i * camNum + j];
    p->correlation /= (camNum * camNum - camNum);
    free(HP);
}

/* patch.cpp:655-721 */
void po_remove_invisible_camera(const po_scene *s, po_patch *p)
{
    if (p->drop) return;
    const int camNum = p->numCam;
    double H[PO_MAX_VIS * 9];
    po_homographies(s, p, p->center, p->normal, H);
    po_set_correlation_table(s, p, H);

    double corrSum;
    double maxCorr = -DBL_MAX;
    int maxIdx = 0;
    for (int i = 0; i < camNum; ++i) {
        corrSum = 0;
        for (int j = 0; j < camNum; ++j) corrSum += p->corrTable[i * camNum + j];
        if (corrSum >= maxCorr) {
            maxIdx = i;
            maxCorr = corrSum;
        }
    }
    double pt[2];
    po_project(s, p->refCamIdx, p->center, pt, p->LOD);

    int removeIdx[PO_MAX_VIS], nrem = 0;
    for (int i = 0; i < camNum; ++i) {
        if (po_region_ratio(s, pt, H + 9 * i) < s->cfg.minRegionRatio) {
            removeIdx[nrem++] = p->camIdx[i];
            continue;
        }
        const po_camera *cam = &s->cams[p->camIdx[i]];
        double neg[3] = {-cam->optN[0], -cam->optN[1], -cam->optN[2]};
        if (dot3(p->normal, neg) < 0) {
            removeIdx[nrem++] = p->camIdx[i];
            continue;
        }
        if (i == maxIdx) continue;
        if (p->corrTable[maxIdx * camNum + i] < s->cfg.minCorrelation) {
            removeIdx[nrem++] = p->camIdx[i];
            continue;
        }
    }
    for (int i = 0; i < nrem; i++) {
        for (int k = 0; k < p->numCam; ++k) {
            if (p->camIdx[k] == removeIdx[i]) {
                for (int q = k; q + 1 < p->numCam; ++q) p->camIdx[q] = p->camIdx[q + 1];
                p->numCam--;
                break;
            }
        }
    }
    if (p->numCam < s->cfg.minCamNum) p->drop = 1;
}

/* patch.cpp:114-176 */
void po_refine(const po_scene *s, po_patch *p)
{
    if (p->numCam < s->cfg.minCamNum) {
        p->fitness = DBL_MAX;
        p->priority = DBL_MAX;
        p->drop = 1;
        return;
    }
    po_set_reference_camera(s, p);
    po_set_depth_and_ray(s, p);
    po_set_depth_range(s, p);
    po_set_lod(s, p);
    if (p->drop) return;

    int beforeRefCamIdx = p->refCamIdx;
    int afterRefCamIdx = -1;
    int beforeCamNum = p->numCam;
    int afterCamNum = -1;
    int count = 0;
    int totalCamNum = beforeCamNum;

    while ((beforeRefCamIdx != afterRefCamIdx || beforeCamNum != afterCamNum) && count++ <= totalCamNum) {
        if (p->numCam < s->cfg.minCamNum) {
            p->fitness = DBL_MAX;
            p->priority = DBL_MAX;
            p->drop = 1;
            return;
        }
        beforeRefCamIdx = p->refCamIdx;
        beforeCamNum = p->numCam;

        po_pso_optimization(s, p);

        if (p->fitness > s->cfg.maxFitness) {
            p->drop = 1;
            return;
        }
        po_remove_invisible_camera(s, p);
        po_set_reference_camera(s, p);
        po_set_depth_and_ray(s, p);
        po_set_depth_range(s, p);
        po_set_lod(s, p);

        if (p->type == PO_TYPE_EXPAND) break;

        afterRefCamIdx = p->refCamIdx;
        afterCamNum = p->numCam;
    }
    po_set_priority(s, p);
    po_set_image_point(s, p);
}

/* MVS::expandCell, mvs.cpp:566-577 (without the insert) */
void po_expand_candidate(const po_scene *s, po_patch *out, const double center[3],
                         const double parentNormal[3], int parentNumCam,
                         const int *parentCamIdx, uint64_t key)
{
    po_patch_init_expand(s, out, center, parentNormal, parentNumCam, parentCamIdx, key);
    po_refine(s, out);
    po_remove_invisible_camera(s, out);
}

/* n independent expansion candidates refined in parallel over PATCHES (one thread per candidate, particles
 * serial) -- not the reference's structure (it parallelises over the <= 30 particles of one patch, patches are
 * sequential): the stronger CPU number bench.py reports next to the reference-structured one. */
void po_expand_candidates_parallel(const po_scene *s, po_patch *out, int n, const double *centers,
                                   const double *parentNormals, const int *parentNumCam, const int *parentCamIdx,
                                   const uint64_t *keys)
{
    po_scene local = *s;
    local.ompParticles = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; ++i)
        po_expand_candidate(&local, &out[i], centers + 3 * i, parentNormals + 3 * i, parentNumCam[i],
                            parentCamIdx + (size_t)PO_MAX_VIS * i, keys[i]);
}

/* mvs.cpp:214-215 */
void po_refine_seed(const po_scene *s, po_patch *p)
{
    po_refine(s, p);
    po_remove_invisible_camera(s, p);
}

/* ------------------------------------------------------------------------ */
/* MVS: mvs/mvs.cpp, mvs/cellmap.cpp                                         */
/* ------------------------------------------------------------------------ */
typedef struct { int n, cap; int *ids; int claimRound; } po_cell;
typedef struct { int width, height; po_cell *cells; } po_cellmap;

struct po_mvs {
    po_scene *s;
    po_patch **patches; /* index == id; NULL once deleted (map<int,Patch>) */
    int nslots, cap;
    int nalive;
    po_cellmap *cellMaps; /* NULL until setCellMaps */
    int *queue;
    int qn, qcap;
    long refineCalls;
    long fitnessEvals;
    int *born;      /* round in which patch id was inserted (-1: before expansion) */
    int bornCap;
    int curRound;
    int thinFront; /* R(B): rounds whose active set has <= thinFront parents take ALL remaining camera slots of every parent */
    long speculative; /* parallel mode: records evaluated ahead that the sequential replay did not consume */
    int parallel;  /* po_mvs_set_parallel: refine() of the units a round is going to look at is evaluated ahead of the sequential
                    * replay, one candidate per OpenMP thread.  refine() is a pure function of (scene, candidate), so the replay
                    * consumes the very same records it would compute itself (tests/test_oracle_parallel.py) */
};

po_mvs *po_mvs_create(po_scene *s)
{
    po_mvs *m = (po_mvs *)calloc(1, sizeof(po_mvs));
    m->s = s;
    m->curRound = -1;
    m->thinFront = PO_DEFAULT_THIN_FRONT;
    return m;
}

static void cellmaps_free(po_mvs *m)
{
    if (!m->cellMaps) return;
    for (int c = 0; c < m->s->numCams; ++c) {
        po_cellmap *cm = &m->cellMaps[c];
        for (long i = 0; i < (long)cm->width * cm->height; ++i) free(cm->cells[i].ids);
        free(cm->cells);
    }
    free(m->cellMaps);
    m->cellMaps = NULL;
}

void po_mvs_destroy(po_mvs *m)
{
    if (!m) return;
    for (int i = 0; i < m->nslots; ++i) free(m->patches[i]);
    free(m->patches);
    cellmaps_free(m);
    free(m->queue);
    free(m->born);
    free(m);
}

static int mvs_store(po_mvs *m, const po_patch *p)
{
    if (m->nslots == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->patches = (po_patch **)realloc(m->patches, sizeof(po_patch *) * (size_t)m->cap);
    }
    po_patch *q = (po_patch *)malloc(sizeof(po_patch));
    *q = *p;
    q->id = m->nslots;
    if (m->nslots >= m->bornCap) {
        m->bornCap = m->bornCap ? m->bornCap * 2 : 1024;
        if (m->bornCap <= m->nslots) m->bornCap = m->nslots + 1024;
        m->born = (int *)realloc(m->born, sizeof(int) * (size_t)m->bornCap);
    }
    m->born[m->nslots] = m->curRound;
    m->patches[m->nslots++] = q;
    m->nalive++;
    return q->id;
}

int po_mvs_add_seed(po_mvs *m, const double center[3], int numCam, const int *camIdx)
{
    po_patch p;
    po_patch_init_seed(m->s, &p, center, numCam, camIdx, (uint64_t)m->nslots);
    return mvs_store(m, &p);
}

void po_svd_solve(int n, int m, const double *A, const double *b, double *x) { svd_solve(n, m, A, b, x); } /* po_seed.c */

/* Patch::reCentering, patch.cpp:67-112 (A.inv(DECOMP_SVD) * b through the SVD back-substitution of the identity) */
void po_recenter(const po_scene *s, int numCam, const int *camIdx, const double *imgPoints, double center[3])
{
    double A[9] = {0}, b[3] = {0, 0, 0};
    for (int i = 0; i < numCam; ++i) {
        const po_camera *cam = &s->cams[camIdx[i]];
        const double p3[3] = {(imgPoints[2 * i] - cam->pp[0]) / cam->focal[0] - cam->T[0],
                              (imgPoints[2 * i + 1] - cam->pp[1]) / cam->focal[1] - cam->T[1], 1.0 - cam->T[2]};
        double w3[3];
        for (int r = 0; r < 3; ++r) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += cam->R[k * 3 + r] * p3[k];
            w3[r] = acc;
        }
        double n[3] = {w3[0] - cam->C[0], w3[1] - cam->C[1], w3[2] - cam->C[2]};
        const double sc = (1.0 / norm3(n));
        for (int k = 0; k < 3; ++k) n[k] = n[k] * sc;
        const double *cc = cam->C;
        A[0] += 1 - n[0] * n[0];
        A[1] += -n[0] * n[1];
        A[2] += -n[0] * n[2];
        A[3] += -n[0] * n[1];
        A[4] += 1 - n[1] * n[1];
        A[5] += -n[1] * n[2];
        A[6] += -n[0] * n[2];
        A[7] += -n[1] * n[2];
        A[8] += 1 - n[2] * n[2];
        b[0] += (1 - n[0] * n[0]) * cc[0] - n[0] * n[1] * cc[1] - n[0] * n[2] * cc[2];
        b[1] += -n[0] * n[1] * cc[0] + (1 - n[1] * n[1]) * cc[1] - n[1] * n[2] * cc[2];
        b[2] += -n[0] * n[2] * cc[0] - n[1] * n[2] * cc[1] + (1 - n[2] * n[2]) * cc[2];
    }
    double inv[9];
    for (int j = 0; j < 3; ++j) {
        double e[3] = {0, 0, 0}, col[3];
        e[j] = 1.0;
        svd_solve(3, 3, A, e, col);
        for (int r = 0; r < 3; ++r) inv[r * 3 + j] = col[r];
    }
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += inv[r * 3 + k] * b[k];
        center[r] = acc;
    }
}

int po_mvs_num_patches(const po_mvs *m) { return m->nalive; }
int po_mvs_num_slots(const po_mvs *m) { return m->nslots; }
const po_patch *po_mvs_get_patch(const po_mvs *m, int id)
{
    if (id < 0 || id >= m->nslots) return NULL;
    return m->patches[id];
}
long po_mvs_refine_calls(const po_mvs *m) { return m->refineCalls; }
long po_mvs_fitness_evals(const po_mvs *m) { return m->fitnessEvals; }

/* mvs.cpp:147-152 + getBoundingVolume :974-997 */
void po_mvs_set_neighbor_radius(po_mvs *m)
{
    double minP[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, maxP[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int id = 0; id < m->nslots; ++id) {
        const po_patch *p = m->patches[id];
        if (!p) continue;
        for (int i = 0; i < 3; ++i) {
            if (p->center[i] < minP[i]) minP[i] = p->center[i];
            if (p->center[i] > maxP[i]) maxP[i] = p->center[i];
        }
    }
    double vol[3] = {maxP[0] - minP[0], maxP[1] - minP[1], maxP[2] - minP[2]};
    double volume = fabs(vol[0] * vol[1] * vol[2]);
    m->s->cfg.neighborRadius = pow(volume, 1.0 / 3.0) * m->s->cfg.neighborRadiusScalar;
}

/* cellmap.cpp:18-23 */
static int cm_in_map(const po_cellmap *cm, int x, int y)
{
    return !(x < 0 || y < 0 || x >= cm->width || y >= cm->height);
}
/* cellmap.cpp:25-29 */
static int cm_insert(po_cellmap *cm, int x, int y, int id)
{
    if (!cm_in_map(cm, x, y)) return 0;
    po_cell *c = &cm->cells[(long)y * cm->width + x];
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 4;
        c->ids = (int *)realloc(c->ids, sizeof(int) * (size_t)c->cap);
    }
    c->ids[c->n++] = id;
    return 1;
}
/* cellmap.cpp:31-38 */
static int cm_drop(po_cellmap *cm, int x, int y, int id)
{
    if (!cm_in_map(cm, x, y)) return 0;
    po_cell *c = &cm->cells[(long)y * cm->width + x];
    for (int i = 0; i < c->n; ++i) {
        if (c->ids[i] == id) {
            for (int k = i; k + 1 < c->n; ++k) c->ids[k] = c->ids[k + 1];
            c->n--;
            return 1;
        }
    }
    return 0;
}

/* mvs.cpp:603-630 */
static void mvs_delete_patch(po_mvs *m, int id)
{
    po_patch *p = (id >= 0 && id < m->nslots) ? m->patches[id] : NULL;
    if (!p) return;
    if (m->cellMaps) {
        const int cellSize = m->s->cfg.cellSize;
        for (int i = 0; i < p->numCam; ++i) {
            int cx = (int)(p->imgPoint[i][0] / cellSize);
            int cy = (int)(p->imgPoint[i][1] / cellSize);
            cm_drop(&m->cellMaps[p->camIdx[i]], cx, cy, p->id);
        }
    }
    free(p);
    m->patches[id] = NULL;
    m->nalive--;
}

/* mvs.cpp:838-898 */
int po_runtime_filtering(const po_mvs *m, const po_patch *pth)
{
    const po_scene *s = m->s;
    const po_config *c = &s->cfg;
    if (pth->drop) return 0;
    if (pth->numCam < c->minCamNum) return 0;
    if (pth->fitness > c->maxFitness) return 0;
    if (pth->fitness == 0.0) return 0;
    if (pth->priority > 10000) return 0;
    if (isnan(pth->fitness)) return 0;
    if (isnan(pth->priority)) return 0;
    if (isnan(pth->correlation)) return 0;
    if (pth->correlation < c->minCorrelation) return 0;

    double pt[2];
    for (int i = 0; i < s->numCams; i++) {
        const po_camera *cam = &s->cams[i];
        if (!po_project(s, i, pth->center, pt, 0)) return 0;
        /* mvs.cpp:860 reads at(cvRound(y), cvRound(x)) after 0 <= pt < dim only: a projection within half a pixel of the
         * right / bottom edge rounds to x == cols / y == rows, an out-of-bounds read in the reference.  Defined here
         * (and in the product's driver) as the edge pixel. */
        int rx = cv_round(pt[0]), ry = cv_round(pt[1]);
        if (rx > cam->width[0] - 1) rx = cam->width[0] - 1;
        if (ry > cam->height[0] - 1) ry = cam->height[0] - 1;
        if (cam->img[0][(size_t)ry * cam->width[0] + rx] == 0) return 0;
    }
    const int camNum = pth->numCam;
    int count = 0;
    for (int i = 0; i < camNum; ++i) {
        const po_camera *cam = &s->cams[pth->camIdx[i]];
        double neg[3] = {-cam->optN[0], -cam->optN[1], -cam->optN[2]};
        if (dot3(pth->normal, neg) > 0) count++;
    }
    if (count < c->minCamNum) return 0;

    if (!m->cellMaps) return 1;
    int fullCellCounter = 0;
    for (int i = 0; i < camNum; ++i) {
        int cx = (int)(pth->imgPoint[i][0] / c->cellSize);
        int cy = (int)(pth->imgPoint[i][1] / c->cellSize);
        const po_cellmap *cm = &m->cellMaps[pth->camIdx[i]];
        /* getCell(cx,cy) has no bounds check in the reference (cellmap.h:26);
         * imgPoint inside every image is guaranteed by the projection test above */
        if (!cm_in_map(cm, cx, cy)) continue;
        const po_cell *cell = &cm->cells[(long)cy * cm->width + cx];
        int found = 0;
        for (int k = 0; k < cell->n; ++k)
            if (cell->ids[k] == pth->id) { found = 1; break; }
        if (found) return 1;
        if (cell->n >= c->maxCellPatchNum && !found) ++fullCellCounter;
    }
    if (fullCellCounter >= camNum) return 0;
    return 1;
}

/* mvs.cpp:196-231 */
void po_mvs_refine_seed_patches(po_mvs *m)
{
    if (m->nalive == 0) return;
    po_mvs_set_neighbor_radius(m);
    char *ahead = NULL; /* parallel mode: seeds whose refine() was evaluated ahead of the sequential loop */
    if (m->parallel) {  /* (nothing a seed's refine() reads changes in that loop) */
        po_scene local = *m->s;
        local.ompParticles = 0;
        ahead = (char *)calloc((size_t)m->nslots + 1, 1);
        for (int id = 0; id < m->nslots; ++id)
            ahead[id] = m->patches[id] && m->patches[id]->numCam >= local.cfg.minCamNum;
#pragma omp parallel for schedule(dynamic, 1)
        for (int id = 0; id < m->nslots; ++id)
            if (ahead[id]) po_refine_seed(&local, m->patches[id]);
    }
    for (int id = 0; id < m->nslots; ++id) {
        po_patch *pth = m->patches[id];
        if (!pth) continue;
        if (ahead ? !ahead[id] : (pth->numCam < m->s->cfg.minCamNum)) { mvs_delete_patch(m, id); continue; }
        if (!ahead) po_refine_seed(m->s, pth);
        m->refineCalls++;
        m->fitnessEvals += pth->psoEvals;
        if (!po_runtime_filtering(m, pth)) { mvs_delete_patch(m, id); continue; }
    }
    free(ahead);
    po_mvs_set_neighbor_radius(m);
}

/* mvs.cpp:74-87, 116-133 ; cellmap.cpp:5-12 */
static void mvs_set_cell_maps(po_mvs *m)
{
    cellmaps_free(m);
    const po_scene *s = m->s;
    m->cellMaps = (po_cellmap *)calloc((size_t)s->numCams, sizeof(po_cellmap));
    for (int c = 0; c < s->numCams; ++c) {
        po_cellmap *cm = &m->cellMaps[c];
        cm->width = cv_ceil((double)s->cams[c].width[0] / (double)s->cfg.cellSize);
        cm->height = cv_ceil((double)s->cams[c].height[0] / (double)s->cfg.cellSize);
        cm->cells = (po_cell *)calloc((size_t)cm->width * cm->height, sizeof(po_cell));
    }
    for (int id = 0; id < m->nslots; ++id) {
        po_patch *pth = m->patches[id];
        if (!pth) continue;
        for (int i = 0; i < pth->numCam; ++i) {
            int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
            int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
            cm_insert(&m->cellMaps[pth->camIdx[i]], cx, cy, pth->id);
        }
    }
}

static void q_push(po_mvs *m, int id)
{
    if (m->qn == m->qcap) {
        m->qcap = m->qcap ? m->qcap * 2 : 1024;
        m->queue = (int *)realloc(m->queue, sizeof(int) * (size_t)m->qcap);
    }
    m->queue[m->qn++] = id;
}
static void q_erase(po_mvs *m, int pos)
{
    memmove(m->queue + pos, m->queue + pos + 1, sizeof(int) * (size_t)(m->qn - pos - 1));
    m->qn--;
}

/* mvs.cpp:656-693 / 695-732 (best: sign=+1, worst: sign=-1) */
static int q_pop_priority(po_mvs *m, int worst)
{
    int top = -1;
    double topPriority = worst ? -DBL_MAX : DBL_MAX;
    /* compacting scan == erase-while-iterating */
    int w = 0;
    for (int r = 0; r < m->qn; ++r) {
        int id = m->queue[r];
        const po_patch *p = po_mvs_get_patch(m, id);
        if (!p) continue;
        if (p->expanded) continue;
        if (worst ? (p->priority > topPriority) : (p->priority < topPriority)) {
            topPriority = p->priority;
            top = w;
        }
        m->queue[w++] = id;
    }
    m->qn = w;
    int topId = -1;
    if (top >= 0) {
        topId = m->queue[top];
        q_erase(m, top);
    }
    return topId;
}
/* mvs.cpp:734-759 */
static int q_pop_breadth(po_mvs *m)
{
    int topId = -1;
    while (m->qn > 0) {
        const po_patch *p = po_mvs_get_patch(m, m->queue[0]);
        if (!p || p->expanded) { q_erase(m, 0); continue; }
        topId = m->queue[0];
        break;
    }
    if (m->qn > 0) q_erase(m, 0); /* queue.erase(it) (UB on an empty queue in the reference) */
    return topId;
}
/* mvs.cpp:761-788 (never examines queue[0]; literal) */
static int q_pop_depth(po_mvs *m)
{
    int topId = -1;
    int it = m->qn - 1;
    while (it > 0) {
        const po_patch *p = po_mvs_get_patch(m, m->queue[it]);
        if (!p || p->expanded) { q_erase(m, it); it = m->qn - 1; continue; }
        topId = m->queue[it];
        break;
    }
    if (it >= 0 && m->qn > 0) q_erase(m, it);
    return topId;
}
/* mvs.cpp:632-654 */
static int q_pop(po_mvs *m)
{
    switch (m->s->cfg.expansionStrategy) {
    default:
    case 0: return q_pop_priority(m, 0);
    case 1: return q_pop_priority(m, 1);
    case 2: return q_pop_breadth(m);
    case 3: return q_pop_depth(m);
    }
}

/* mvs.cpp:792-807.  beforeRound >= 0: evaluate on the state before that round started
 * (patches inserted during it are ignored) -- used by the cell-claim rule of R(B). */
static int skip_neighbor_cell(const po_mvs *m, const po_cell *cell, const po_patch *refPth, int beforeRound)
{
    int pthNum = 0;
    for (int k = 0; k < cell->n; k++)
        if (beforeRound < 0 || m->born[cell->ids[k]] < beforeRound) pthNum++;
    if (pthNum >= m->s->cfg.maxCellPatchNum) return 1;
    for (int k = 0; k < cell->n; k++) {
        if (beforeRound >= 0 && m->born[cell->ids[k]] >= beforeRound) continue;
        const po_patch *pth = po_mvs_get_patch(m, cell->ids[k]);
        if (!pth) continue;
        if (pth->correlation > m->s->cfg.minCorrelation) return 1;
        if (po_is_neighbor(m->s, refPth, pth)) return 1;
    }
    return 0;
}

/* mvs.cpp:809-836 */
void po_expansion_center(const po_scene *s, int camI, const po_patch *parent, int cx, int cy, double center[3])
{
    const po_camera *cam = &s->cams[camI];
    const double px = (cx + 0.5) * s->cfg.cellSize;
    const double py = (cy + 0.5) * s->cfg.cellSize;
    double p3d[3], tmp[3];
    p3d[0] = (px - cam->pp[0]) / cam->focal[0];
    p3d[1] = (py - cam->pp[1]) / cam->focal[1];
    p3d[2] = 1.0;
    for (int i = 0; i < 3; ++i) tmp[i] = p3d[i] - cam->T[i];
    /* R^T * tmp : gemm with GEMM_1_T, k = 0..2 */
    for (int i = 0; i < 3; ++i) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += cam->R[k * 3 + i] * tmp[k];
        p3d[i] = a;
    }
    double v13[3], v12[3];
    for (int i = 0; i < 3; ++i) v13[i] = parent->center[i] - cam->C[i];
    for (int i = 0; i < 3; ++i) v12[i] = p3d[i] - cam->C[i];
    const double u = dot3(parent->normal, v13) / dot3(parent->normal, v12);
    for (int i = 0; i < 3; ++i) center[i] = cam->C[i] + u * v12[i];
}

/* mvs.cpp:579-601 */
static void mvs_insert_patch(po_mvs *m, po_patch *pth)
{
    pth->id = m->nslots; /* id the patch will get; used by runtimeFiltering's find() */
    if (!po_runtime_filtering(m, pth)) return;
    int id = mvs_store(m, pth);
    q_push(m, id);
    const int cellSize = m->s->cfg.cellSize;
    for (int i = 0; i < pth->numCam; ++i) {
        int cx = (int)(pth->imgPoint[i][0] / cellSize);
        int cy = (int)(pth->imgPoint[i][1] / cellSize);
        cm_insert(&m->cellMaps[pth->camIdx[i]], cx, cy, id);
    }
}

/* expandCell (mvs.cpp:566-577) for neighbour j of camera slot i of a parent, without the insert */
static void mvs_make_child(const po_scene *s, const po_patch *pth, int i, int j, po_patch *child)
{
    const int camI = pth->camIdx[i];
    int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
    int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
    const int nx[] = {cx - 1, cx, cx + 1, cx};
    const int ny[] = {cy, cy - 1, cy, cy + 1};
    double center[3];
    po_expansion_center(s, camI, pth, nx[j], ny[j], center);
    po_expand_candidate(s, child, center, pth->normal, pth->numCam, pth->camIdx, po_child_key(pth->key, camI, nx[j], ny[j]));
}
/* expandCell (mvs.cpp:566-577) for neighbour j of camera slot i of a parent */
static void mvs_expand_one(po_mvs *m, int parentId, int i, int j)
{
    const po_scene *s = m->s;
    const po_patch *pth = m->patches[parentId];
    const int camI = pth->camIdx[i];
    int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
    int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
    const int nx[] = {cx - 1, cx, cx + 1, cx};
    const int ny[] = {cy, cy - 1, cy, cy + 1};
    double center[3];
    po_expansion_center(s, camI, pth, nx[j], ny[j], center);
    po_patch child;
    po_expand_candidate(s, &child, center, pth->normal, pth->numCam, pth->camIdx,
                        po_child_key(pth->key, camI, nx[j], ny[j]));
    m->refineCalls++;
    m->fitnessEvals += child.psoEvals;
    mvs_insert_patch(m, &child);
}

typedef struct { int id, slot, j; } po_unit;

/* MVS::expansionPatches (mvs.cpp:233-275) generalised to the rounds R(B) of DESIGN.md section 6:
 *   - an ordered active set holds up to B popped parents, each with a camera-slot cursor; a round
 *     first tops the set up from the queue (reference pop policy, setExpanded,
 *     runtimeFiltering/delete, mvs.cpp:245-260);
 *   - the round's work list is [units deferred by the previous round] followed by the <= 4
 *     neighbour cells of the CURRENT camera slot of every active parent, in activation order
 *     (the body of expandNeighborCell's outer loop, mvs.cpp:535-563);
 *   - a unit whose target cell is already blocked on the state BEFORE the round is dropped
 *     (that is skipNeighborCell, :558); of several units of one round that target the same
 *     (camera, cell) only the first is handled now -- the cell is "claimed" -- the others are
 *     deferred to the head of the next round;
 *   - a handled unit re-applies skipNeighborCell on the live state, then expandCell + insertPatch;
 *   - cursors advance; a parent leaves the set after its last visible camera.
 * With B = 1 no two units of a round share a cell, nothing is ever deferred and the sequence of
 * refine()/insertPatch() calls is the reference's own (including its quirk that the parent popped
 * last is not expanded, :241-243,271). */
long po_mvs_expansion_patches(po_mvs *m, int B, int maxRounds, int strictTail)
{
    long before = m->refineCalls;
    const po_scene *s = m->s;
    mvs_set_cell_maps(m);
    m->qn = 0;
    for (int id = 0; id < m->nslots; ++id)
        if (m->patches[id]) q_push(m, id); /* initPriorityQueue :89-95 */
    po_mvs_set_neighbor_radius(m);
    if (B < 1) B = 1;
    int *actId = (int *)malloc(sizeof(int) * (size_t)B);
    int *actSlot = (int *)malloc(sizeof(int) * (size_t)B);
    int nA = 0, rounds = 0, stop = 0;
    po_unit *deferred = NULL, *nextDef = NULL, *work = NULL;
    int nDef = 0, nNext = 0, capDef = 0, capNext = 0, capWork = 0;
    uint64_t *claims = NULL;
    int capClaims = 0;
    m->curRound = 0;
    for (;;) {
        while (nA < B && !stop) {
            int id = q_pop(m);
            if (id < 0) break;
            if (strictTail && B == 1 && m->qn == 0) { stop = 1; break; }
            po_patch *pth = m->patches[id];
            pth->expanded = 1;
            if (!po_runtime_filtering(m, pth)) { mvs_delete_patch(m, id); continue; }
            actId[nA] = id;
            actSlot[nA] = 0;
            nA++;
        }
        if (nA == 0 && nDef == 0) break;
        /* work list of the round: one camera slot of every active parent; a thin front (few active
         * parents -- the long tail of the expansion) takes all remaining slots of its parents at once */
        const int thin = nA <= m->thinFront;
        int nW = nDef;
        for (int a = 0; a < nA; ++a) nW += 4 * (thin ? m->patches[actId[a]]->numCam - actSlot[a] : 1);
        if (nW > capWork) { capWork = nW * 2; work = (po_unit *)realloc(work, sizeof(po_unit) * (size_t)capWork); }
        nW = 0;
        for (int d = 0; d < nDef; ++d) work[nW++] = deferred[d];
        for (int a = 0; a < nA; ++a) {
            const int sEnd = thin ? m->patches[actId[a]]->numCam : actSlot[a] + 1;
            for (int sl = actSlot[a]; sl < sEnd; ++sl)
                for (int j = 0; j < 4; ++j) { work[nW].id = actId[a]; work[nW].slot = sl; work[nW].j = j; nW++; }
            actSlot[a] = sEnd - 1; /* advanced past sEnd below */
        }
        if (nW > capClaims) { capClaims = nW * 2; claims = (uint64_t *)realloc(claims, sizeof(uint64_t) * (size_t)capClaims); }
        int nClaims = 0;
        nNext = 0;
        /* po_mvs_set_parallel: the same walk in three phases.  (A) what a unit does -- dropped, deferred or claimed -- depends on
         * the state before the round alone (skip test with beforeRound, claims in work-list order); (B) refine() of the
         * claimed units, one per thread; (C) the sequential replay of the loop below with the records of (B). */
        if (m->parallel) {
            int *cl = (int *)malloc(sizeof(int) * (size_t)(nW > 0 ? nW : 1));
            int nCl = 0;
            for (int u = 0; u < nW; ++u) {
                const po_patch *pth = m->patches[work[u].id];
                const int i = work[u].slot, j = work[u].j;
                const int camI = pth->camIdx[i];
                po_cellmap *map = &m->cellMaps[camI];
                int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
                int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
                const int nx[] = {cx - 1, cx, cx + 1, cx};
                const int ny[] = {cy, cy - 1, cy, cy + 1};
                if (!cm_in_map(map, nx[j], ny[j])) continue;
                po_cell *cell = &map->cells[(long)ny[j] * map->width + nx[j]];
                if (skip_neighbor_cell(m, cell, pth, m->curRound)) continue;
                if (cell->claimRound == m->curRound + 1) { /* (stamp = round + 1: calloc'ed cells start unclaimed) */
                    if (nNext == capNext) { capNext = capNext ? capNext * 2 : 256; nextDef = (po_unit *)realloc(nextDef, sizeof(po_unit) * (size_t)capNext); }
                    nextDef[nNext++] = work[u];
                    continue;
                }
                cell->claimRound = m->curRound + 1;
                cl[nCl++] = u;
            }
            po_patch *pre = (po_patch *)malloc(sizeof(po_patch) * (size_t)(nCl > 0 ? nCl : 1));
            {
                po_scene local = *s;
                local.ompParticles = 0;
#pragma omp parallel for schedule(dynamic, 1)
                for (int q = 0; q < nCl; ++q)
                    mvs_make_child(&local, m->patches[work[cl[q]].id], work[cl[q]].slot, work[cl[q]].j, &pre[q]);
            }
            for (int q = 0; q < nCl; ++q) {
                const int u = cl[q];
                const po_patch *pth = m->patches[work[u].id];
                const int i = work[u].slot, j = work[u].j;
                const int camI = pth->camIdx[i];
                po_cellmap *map = &m->cellMaps[camI];
                int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
                int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
                const int nx[] = {cx - 1, cx, cx + 1, cx};
                const int ny[] = {cy, cy - 1, cy, cy + 1};
                const po_cell *cell = &map->cells[(long)ny[j] * map->width + nx[j]];
                if (skip_neighbor_cell(m, cell, pth, -1)) continue; /* mvs.cpp:558 on the live state */
                m->refineCalls++;
                m->fitnessEvals += pre[q].psoEvals;
                m->speculative--; /* (counted below: claimed units whose record the replay did not consume) */
                mvs_insert_patch(m, &pre[q]);
            }
            m->speculative += nCl;
            free(pre);
            free(cl);
        }
        for (int u = 0; u < nW && !m->parallel; ++u) {
            const po_patch *pth = m->patches[work[u].id];
            const int i = work[u].slot, j = work[u].j;
            const int camI = pth->camIdx[i];
            po_cellmap *map = &m->cellMaps[camI];
            int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize);
            int cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
            const int nx[] = {cx - 1, cx, cx + 1, cx};
            const int ny[] = {cy, cy - 1, cy, cy + 1};
            if (!cm_in_map(map, nx[j], ny[j])) continue;
            const po_cell *cell = &map->cells[(long)ny[j] * map->width + nx[j]];
            if (skip_neighbor_cell(m, cell, pth, m->curRound)) continue; /* blocked before the round */
            const uint64_t key = (((uint64_t)(uint32_t)camI) << 48) ^ (((uint64_t)(uint32_t)nx[j]) << 24) ^ (uint64_t)(uint32_t)ny[j];
            int taken = 0;
            for (int q = 0; q < nClaims; ++q)
                if (claims[q] == key) { taken = 1; break; }
            if (taken) { /* a cell takes one attempt per round: retry next round */
                if (nNext == capNext) { capNext = capNext ? capNext * 2 : 256; nextDef = (po_unit *)realloc(nextDef, sizeof(po_unit) * (size_t)capNext); }
                nextDef[nNext++] = work[u];
                continue;
            }
            claims[nClaims++] = key;
            if (skip_neighbor_cell(m, cell, pth, -1)) continue; /* mvs.cpp:558 on the live state */
            mvs_expand_one(m, work[u].id, i, j);
        }
        /* advance */
        for (int a = 0; a < nA; ++a) actSlot[a]++;
        int w = 0;
        for (int a = 0; a < nA; ++a) {
            if (actSlot[a] < m->patches[actId[a]]->numCam) {
                actId[w] = actId[a];
                actSlot[w] = actSlot[a];
                w++;
            }
        }
        nA = w;
        { po_unit *t = deferred; deferred = nextDef; nextDef = t; int c = capDef; capDef = capNext; capNext = c; }
        nDef = nNext;
        rounds++;
        m->curRound++;
        if (maxRounds > 0 && rounds >= maxRounds) break;
    }
    free(actId);
    free(actSlot);
    free(deferred);
    free(nextDef);
    free(work);
    free(claims);
    m->curRound = -1;
    po_mvs_set_neighbor_radius(m);
    return m->refineCalls - before;
}

/* ------------------------------------------------------------------------ */
/* post filters: the `-f` verb, TMVS.cpp:124-172                             */
/* ------------------------------------------------------------------------ */
/* Patch(center, normalS, camIdx, fitness, correlation, id), patch.cpp:45-59 (FileLoader::loadMvsPatch) */
int po_mvs_load_patch(po_mvs *m, const double center[3], const double normalS[2], int numCam, const int *camIdx,
                      double fitness, double correlation)
{
    const po_scene *s = m->s;
    po_patch p;
    patch_init(&p);
    p.type = PO_TYPE_SEED;
    for (int i = 0; i < 3; ++i) p.center[i] = center[i];
    p.numCam = numCam > PO_MAX_VIS ? PO_MAX_VIS : numCam;
    for (int i = 0; i < p.numCam; ++i) p.camIdx[i] = camIdx[i];
    p.fitness = fitness;
    p.correlation = correlation;
    p.drop = 0;
    p.key = (uint64_t)m->nslots;
    p.normalS[0] = normalS[0];
    p.normalS[1] = normalS[1];
    s2n(s, normalS, p.normal); /* setNormal(Vec2d), abstractpatch.cpp:48-51 */
    po_set_reference_camera(s, &p);
    po_set_depth_and_ray(s, &p);
    po_set_depth_range(s, &p);
    po_set_lod(s, &p);
    po_set_priority(s, &p);
    po_set_image_point(s, &p);
    p.drop = 0; /* the filter verbs never look at it; the loader does not test it either */
    int id = mvs_store(m, &p);
    m->patches[id]->expanded = 1;
    return id;
}

static void filter_prepare(po_mvs *m)
{
    if (!m->cellMaps) { /* `if (cellMaps.empty())` at the head of every filter */
        po_mvs_set_neighbor_radius(m);
        mvs_set_cell_maps(m);
    }
}

/* MVS::cellFiltering, mvs.cpp:278-325 */
void po_mvs_cell_filtering(po_mvs *m)
{
    filter_prepare(m);
    const po_scene *s = m->s;
    for (int ci = 0; ci < s->numCams; ++ci) {
        po_cellmap *map = &m->cellMaps[ci];
        for (int x = 0; x < map->width; ++x) {
            for (int y = 0; y < map->height; ++y) {
                po_cell *cell = &map->cells[(long)y * map->width + x];
                const int pthNum = cell->n;
                if (pthNum == 0) continue;
                int *removeIdx = (int *)malloc(sizeof(int) * (size_t)pthNum);
                int nRem = 0;
                for (int j = 0; j < pthNum; ++j) {
                    double corrSum = 0;
                    for (int k = 0; k < pthNum; ++k) {
                        if (j == k) continue;
                        const po_patch *q = m->patches[cell->ids[k]];
                        if (!q) continue;
                        corrSum += q->correlation;
                    }
                    const po_patch *pth = m->patches[cell->ids[j]];
                    if (!pth) continue;
                    if (pth->correlation * pth->numCam < corrSum) removeIdx[nRem++] = cell->ids[j];
                }
                for (int j = 0; j < nRem; ++j) mvs_delete_patch(m, removeIdx[j]);
                free(removeIdx);
            }
        }
    }
}

/* MVS::visibilityFiltering, mvs.cpp:394-446 */
void po_mvs_visibility_filtering(po_mvs *m)
{
    filter_prepare(m);
    const po_scene *s = m->s;
    for (int id = 0; id < m->nslots; ++id) {
        const po_patch *pth = m->patches[id];
        if (!pth) continue;
        int visibleCount = pth->numCam;
        for (int i = 0; i < pth->numCam; ++i) {
            const po_camera *cam = &s->cams[pth->camIdx[i]];
            double d[3] = {pth->center[0] - cam->C[0], pth->center[1] - cam->C[1], pth->center[2] - cam->C[2]};
            const double depth = norm3(d);
            const int cx = (int)(pth->imgPoint[i][0] / s->cfg.cellSize), cy = (int)(pth->imgPoint[i][1] / s->cfg.cellSize);
            po_cellmap *map = &m->cellMaps[pth->camIdx[i]];
            if (!cm_in_map(map, cx, cy)) continue; /* undefined in the reference; does not happen for loaded clouds */
            const po_cell *cell = &map->cells[(long)cy * map->width + cx];
            for (int k = 0; k < cell->n; ++k) {
                if (cell->ids[k] == pth->id) continue;
                const po_patch *q = m->patches[cell->ids[k]];
                if (!q) continue;
                double dn[3] = {q->center[0] - cam->C[0], q->center[1] - cam->C[1], q->center[2] - cam->C[2]};
                if (depth > norm3(dn)) {
                    --visibleCount;
                    break;
                }
            }
        }
        if (visibleCount < s->cfg.minCamNum) mvs_delete_patch(m, id);
    }
}

/* MVS::neighborCellFiltering, mvs.cpp:327-392 */
void po_mvs_neighbor_cell_filtering(po_mvs *m, double neighborRatio)
{
    filter_prepare(m);
    const po_scene *s = m->s;
    for (int ci = 0; ci < s->numCams; ++ci) {
        po_cellmap *map = &m->cellMaps[ci];
        for (int x = 0; x < map->width; ++x) {
            for (int y = 0; y < map->height; ++y) {
                po_cell *cell = &map->cells[(long)y * map->width + x];
                const int pthNum = cell->n;
                if (pthNum == 0) continue;
                int *removeIdx = (int *)malloc(sizeof(int) * (size_t)pthNum);
                int nRem = 0;
                const int nx[9] = {x, x - 1, x + 1, x - 1, x + 1, x + 1, x, x - 1, x};
                const int ny[9] = {y, y - 1, y - 1, y + 1, y + 1, y, y + 1, y, y - 1};
                for (int j = 0; j < pthNum; ++j) {
                    const po_patch *c = m->patches[cell->ids[j]];
                    if (!c) continue;
                    int neighborPthSum = 0, neighborPthNum = 0;
                    for (int q = 0; q < 9; ++q) {
                        if (!cm_in_map(map, nx[q], ny[q])) continue;
                        const po_cell *nc = &map->cells[(long)ny[q] * map->width + nx[q]];
                        neighborPthSum += nc->n;
                        for (int k = 0; k < nc->n; ++k) {
                            const po_patch *np = m->patches[nc->ids[k]];
                            if (!np) continue;
                            if (po_is_neighbor(s, c, np)) ++neighborPthNum;
                        }
                    }
                    if ((double)neighborPthNum / (double)neighborPthSum < neighborRatio) removeIdx[nRem++] = c->id;
                }
                for (int j = 0; j < nRem; ++j) mvs_delete_patch(m, removeIdx[j]);
                free(removeIdx);
            }
        }
    }
}

/* MVS::neighborPatchFiltering, mvs.cpp:448-524.  The reference sorts every patch's distances to all others and walks
 * them up to the first one beyond neighborRadius: that is the count of the others within the radius.  counts (optional):
 * nslots ints, -1 for deleted ids. */
void po_mvs_neighbor_patch_filtering(po_mvs *m, double neighborRatio, int *counts)
{
    filter_prepare(m);
    const double radius = m->s->cfg.neighborRadius;
    int n = 0;
    int *cnt = (int *)malloc(sizeof(int) * (size_t)(m->nslots > 0 ? m->nslots : 1));
    for (int a = 0; a < m->nslots; ++a) {
        cnt[a] = -1;
        const po_patch *p = m->patches[a];
        if (!p) continue;
        int c = 0;
        for (int b = 0; b < m->nslots; ++b) {
            const po_patch *q = m->patches[b];
            if (!q || b == a) continue;
            double d[3] = {p->center[0] - q->center[0], p->center[1] - q->center[1], p->center[2] - q->center[2]};
            if (!(norm3(d) > radius)) ++c;
        }
        cnt[a] = c;
        ++n;
    }
    double avg = 0;
    for (int a = 0; a < m->nslots; ++a)
        if (cnt[a] >= 0) avg += (double)cnt[a];
    if (n > 0) avg /= (double)n;
    if (counts) memcpy(counts, cnt, sizeof(int) * (size_t)m->nslots);
    for (int a = 0; a < m->nslots; ++a)
        if (cnt[a] >= 0 && (double)cnt[a] < (avg * neighborRatio)) mvs_delete_patch(m, a);
    free(cnt);
}

void po_mvs_set_thin_front(po_mvs *m, int thinFront) { m->thinFront = thinFront < 0 ? 0 : thinFront; }
void po_mvs_set_parallel(po_mvs *m, int on) { m->parallel = on ? 1 : 0; }
long po_mvs_speculative(const po_mvs *m) { return m->speculative; }

size_t po_sizeof_patch(void) { return sizeof(po_patch); }
size_t po_sizeof_config(void) { return sizeof(po_config); }
size_t po_sizeof_camera(void) { return sizeof(po_camera); }
