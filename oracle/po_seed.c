/* TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of FeatureManager::setSeedPatches after its SIFT call
 * (mvs/featuremanager.cpp:28-99, 118-287): brute-force cross-checked descriptor matching, epipolar-line filtering with
 * the fundamental matrices of the camera pairs, removal of non-cross matches and of weakly matched views, union of the
 * pairwise matches into n-view features, one seed per feature through Patch::reCentering.
 *
 * Parity: UNPINNED (OpenCV's SIFT / BFMatcher / Mat::inv are not in this image).  What is restated:
 *  - BFMatcher(NORM_L2, crossCheck = true).match(query, train): for every query descriptor the train descriptor of
 *    least L2 distance (first minimum), kept when that train descriptor's own nearest query is the same one.  The
 *    distance is evaluated in float in the order of OpenCV 2.4's portable normL2Sqr_ loop (four differences per step,
 *    s += v0*v0 + v1*v1 + v2*v2 + v3*v3), square root in float; matches come out in query order.
 *  - everything else is list logic and small double-precision algebra stated as in the source.
 * Only tests/ may use this file. */
#include "pais_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* pais_oracle.c */
void po_svd_solve(int n, int m, const double *A, const double *b, double *x);

/* FeatureManager::getFundamental, featuremanager.cpp:245-262: F = [eT]x * pT * pF^+, eT = pT * (cF, 1) */
void po_seed_fundamental(const po_camera *from, const po_camera *to, double F[9])
{
    double pF[12], pT[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            pF[r * 4 + c] = from->KR[r * 3 + c];
            pT[r * 4 + c] = to->KR[r * 3 + c];
        }
        pF[r * 4 + 3] = from->KT[r];
        pT[r * 4 + 3] = to->KT[r];
    }
    const double cF[4] = {from->C[0], from->C[1], from->C[2], 1.0};
    double eT[3];
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += pT[r * 4 + k] * cF[k];
        eT[r] = acc;
    }
    const double ex[9] = {0, -eT[2], eT[1], eT[2], 0, -eT[0], -eT[1], eT[0], 0};
    /* pF.inv(DECOMP_SVD) of the 3 x 4 matrix: the pseudo-inverse, through the least-squares solves of pF^T x = e_k */
    double pFt[12]; /* 4 x 3 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) pFt[c * 3 + r] = pF[r * 4 + c];
    double pinv[12]; /* 4 x 3: pinv[k][r] = x_k[r] */
    for (int k = 0; k < 4; ++k) {
        double e[4] = {0, 0, 0, 0}, x[3];
        e[k] = 1.0;
        po_svd_solve(4, 3, pFt, e, x);
        for (int r = 0; r < 3; ++r) pinv[k * 3 + r] = x[r];
    }
    double M[12]; /* exT * pT : 3 x 4 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += ex[r * 3 + k] * pT[k * 4 + c];
            M[r * 4 + c] = acc;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += M[r * 4 + k] * pinv[k * 3 + c];
            F[r * 3 + c] = acc;
        }
}

/* getFundamentalMatrices, :265-287: M[i][j] = F(from j, to i) for i < j, M[j][i] = M[i][j]^T, identity on the diagonal */
void po_seed_fundamentals(const po_scene *s, double *Fs /* numCams x numCams x 9 */)
{
    const int n = s->numCams;
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double *Fij = Fs + ((size_t)i * n + j) * 9, *Fji = Fs + ((size_t)j * n + i) * 9;
            if (i == j) {
                for (int k = 0; k < 9; ++k) Fij[k] = (k % 4 == 0) ? 1.0 : 0.0;
            } else {
                po_seed_fundamental(&s->cams[j], &s->cams[i], Fij);
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) Fji[r * 3 + c] = Fij[c * 3 + r];
            }
        }
}

static float l2_float(const float *a, const float *b, int n)
{
    float s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const float v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1], v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; ++i) {
        const float v = a[i] - b[i];
        s += v * v;
    }
    return sqrtf(s);
}

/* nearest train descriptor of every query descriptor (first minimum) */
static void nearest(int nq, const float *dq, int nt, const float *dt, int dim, int *best, float *dist)
{
    for (int q = 0; q < nq; ++q) {
        int b = -1;
        float bd = FLT_MAX;
        for (int t = 0; t < nt; ++t) {
            const float d = l2_float(dq + (size_t)q * dim, dt + (size_t)t * dim, dim);
            if (d < bd) { bd = d; b = t; }
        }
        best[q] = b;
        dist[q] = bd;
    }
}

/* BFMatcher(NORM_L2, true).match: train_of_query[q] = t or -1 */
void po_seed_match(int nq, const float *dq, int nt, const float *dt, int dim, int *train_of_query, float *dist)
{
    int *bq = (int *)malloc(sizeof(int) * (size_t)(nq > 0 ? nq : 1)), *bt = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    float *dd = (float *)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    nearest(nq, dq, nt, dt, dim, bq, dist);
    nearest(nt, dt, nq, dq, dim, bt, dd);
    for (int q = 0; q < nq; ++q) train_of_query[q] = (bq[q] >= 0 && bt[bq[q]] == q) ? bq[q] : -1;
    free(bq); free(bt); free(dd);
}

typedef struct { int q, t; } Match;
typedef struct { Match *v; int n, cap; } MatchList;
static void ml_push(MatchList *l, Match m)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 16; l->v = (Match *)realloc(l->v, sizeof(Match) * (size_t)l->cap); }
    l->v[l->n++] = m;
}
static void ml_erase(MatchList *l, int i)
{
    memmove(l->v + i, l->v + i + 1, sizeof(Match) * (size_t)(l->n - i - 1));
    l->n--;
}

typedef struct { int cam, feat; } NV;
typedef struct { NV *v; int n, cap; } NVList;
static void nv_push(NVList *l, NV e)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 4; l->v = (NV *)realloc(l->v, sizeof(NV) * (size_t)l->cap); }
    l->v[l->n++] = e;
}

/* setSeedPatches, :28-99, from the keypoints and descriptors on.  kp_n[c], kp_xy[c] (n x 2 floats), kp_desc[c] (n x dim).
 * Output: *out_feat = malloc'ed list of n-view features as (numCam, then numCam x (cam, feat)) ints in creation order,
 * restricted to those with >= minCamNum views; *out_centers = their re-centred 3-D points.  Returns their number. */
int po_seed_features(const po_scene *s, const int *kp_n, const float *const *kp_xy, const float *const *kp_desc, int dim,
                     double maxDist, int **out_feat, int *out_feat_len, double **out_centers)
{
    const int C = s->numCams;
    double *Fs = (double *)malloc(sizeof(double) * 9 * (size_t)C * C);
    po_seed_fundamentals(s, Fs);
    MatchList *table = (MatchList *)calloc((size_t)C * C, sizeof(MatchList));
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            if (i == j) continue;
            int *tq = (int *)malloc(sizeof(int) * (size_t)(kp_n[i] > 0 ? kp_n[i] : 1));
            float *dd = (float *)malloc(sizeof(float) * (size_t)(kp_n[i] > 0 ? kp_n[i] : 1));
            po_seed_match(kp_n[i], kp_desc[i], kp_n[j], kp_desc[j], dim, tq, dd);
            const double *F = Fs + ((size_t)i * C + j) * 9;
            for (int q = 0; q < kp_n[i]; ++q) {
                if (tq[q] < 0) continue;
                /* epipolarLineFiltering, :158-196: epiLine = q^T F, distance of t to it */
                const double qx = kp_xy[i][2 * q], qy = kp_xy[i][2 * q + 1];
                const double tx = kp_xy[j][2 * tq[q]], ty = kp_xy[j][2 * tq[q] + 1];
                double l[3];
                for (int c = 0; c < 3; ++c) l[c] = qx * F[0 * 3 + c] + qy * F[1 * 3 + c] + 1.0 * F[2 * 3 + c];
                const double d = fabs(l[0] * tx + l[1] * ty + l[2] * 1.0) / sqrt(l[0] * l[0] + l[1] * l[1]);
                if (d > maxDist) continue;
                Match m = {q, tq[q]};
                ml_push(&table[(size_t)i * C + j], m);
            }
            free(tq); free(dd);
        }
    /* filteroutNonMatches, :198-243 */
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            MatchList *a = &table[(size_t)i * C + j], *b = &table[(size_t)j * C + i];
            for (int it = 0; it < a->n;) {
                int cross = 0;
                for (int it2 = 0; it2 < b->n; ++it2)
                    if (a->v[it].t == b->v[it2].q && a->v[it].q == b->v[it2].t) {
                        cross = 1;
                        ml_erase(b, it2);
                        break;
                    }
                if (!cross) { ml_erase(a, it); continue; }
                ++it;
            }
        }
    for (int i = 0; i < C; ++i) {
        int maxMatch = 0;
        for (int j = 0; j < C; ++j)
            if (table[(size_t)i * C + j].n > maxMatch) maxMatch = table[(size_t)i * C + j].n;
        for (int j = 0; j < C; ++j)
            if (table[(size_t)i * C + j].n < maxMatch / 4.0) table[(size_t)i * C + j].n = 0;
    }
    /* union, :56-82 with setNVMatch :118-156 */
    NVList *nv = NULL;
    int nnv = 0, capnv = 0;
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            if (i == j) continue;
            MatchList *l = &table[(size_t)i * C + j];
            while (l->n > 0) {
                const Match m = l->v[l->n - 1];
                int found = 0;
                for (int u = 0; u < nnv && !found; ++u) {
                    NVList *f = &nv[u];
                    for (int e = 0; e < f->n; ++e) {
                        if (i == f->v[e].cam && m.q == f->v[e].feat) {
                            int k = 0;
                            for (; k < f->n; ++k)
                                if (f->v[k].feat == m.t) break; /* the source compares the feature index only */
                            if (k == f->n) { NV ne = {j, m.t}; nv_push(f, ne); }
                            found = 1;
                            break;
                        }
                        if (j == f->v[e].cam && m.t == f->v[e].feat) {
                            int k = 0;
                            for (; k < f->n; ++k)
                                if (f->v[k].feat == m.q) break;
                            if (k == f->n) { NV ne = {i, m.q}; nv_push(f, ne); }
                            found = 1;
                            break;
                        }
                    }
                }
                if (!found) {
                    if (nnv == capnv) { capnv = capnv ? 2 * capnv : 64; nv = (NVList *)realloc(nv, sizeof(NVList) * (size_t)capnv); }
                    NVList f = {NULL, 0, 0};
                    NV a = {i, m.q}, b = {j, m.t};
                    nv_push(&f, a);
                    nv_push(&f, b);
                    nv[nnv++] = f;
                }
                l->n--;
            }
        }
    /* seeds, :84-99 */
    int total = 0, kept = 0;
    for (int u = 0; u < nnv; ++u)
        if (nv[u].n >= s->cfg.minCamNum) { total += 1 + 2 * nv[u].n; kept++; }
    int *feat = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
    double *cen = (double *)malloc(sizeof(double) * 3 * (size_t)(kept > 0 ? kept : 1));
    int w = 0, kk = 0;
    for (int u = 0; u < nnv; ++u) {
        if (nv[u].n < s->cfg.minCamNum) continue;
        feat[w++] = nv[u].n;
        int *cams = (int *)malloc(sizeof(int) * (size_t)nv[u].n);
        double *pts = (double *)malloc(sizeof(double) * 2 * (size_t)nv[u].n);
        for (int e = 0; e < nv[u].n; ++e) {
            feat[w++] = nv[u].v[e].cam;
            feat[w++] = nv[u].v[e].feat;
            cams[e] = nv[u].v[e].cam;
            pts[2 * e] = (double)kp_xy[nv[u].v[e].cam][2 * nv[u].v[e].feat];
            pts[2 * e + 1] = (double)kp_xy[nv[u].v[e].cam][2 * nv[u].v[e].feat + 1];
        }
        po_recenter(s, nv[u].n, cams, pts, cen + 3 * kk);
        free(cams); free(pts);
        kk++;
    }
    for (int u = 0; u < nnv; ++u) free(nv[u].v);
    free(nv);
    for (int k = 0; k < C * C; ++k) free(table[k].v);
    free(table);
    free(Fs);
    *out_feat = feat;
    *out_feat_len = w;
    *out_centers = cen;
    return kept;
}
void po_seed_free(void *p) { free(p); }
