// pais_seed.hip -- the heavy part of FeatureManager::setSeedPatches (mvs/featuremanager.cpp:28-39): brute-force nearest
// descriptor search on the GPU, and the fundamental matrix of a camera pair (:245-262).  The list logic that follows
// (filters, union, seeds) lives with the driver in pais_mvs.hip.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <string>
#include <vector>
#include "../../include/pais_seed.h"
#include "pais_dev.hpp"

static thread_local std::string g_seed_err;
extern "C" const char *pais_seed_last_error(void) { return g_seed_err.c_str(); }
static int sfail(const char *m) { g_seed_err = m; return -1; }
#define SHIP(call)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) { g_seed_err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } \
    } while (0)

// One wave per query descriptor: the query sits in LDS, lane l walks the train descriptors l, l + 64, ... and keeps its
// first minimum; the wave then takes the least distance, smallest index on ties -- the first minimum of a sequential scan.
// The distance is evaluated in float, four differences per step as ((v0 v0 + v1 v1) + v2 v2) + v3 v3 added to the running
// sum, square root in float (cv::normL2Sqr_'s portable loop + BFMatcher's sqrt; no contraction: -ffp-contract=off).
__global__ __launch_bounds__(64) void k_nearest_descriptor(const float *query, int nq, const float *train, int nt, int dim,
                                                           int32_t *best, float *bestDist)
{
    extern __shared__ float qs[];
    const int lane = threadIdx.x;
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        __syncthreads();
        for (int i = lane; i < dim; i += 64) qs[i] = query[(size_t)q * dim + i];
        __syncthreads();
        float bd = FLT_MAX;
        int bi = -1;
        for (int t = lane; t < nt; t += 64) {
            const float *b = train + (size_t)t * dim;
            float s = 0;
            int i = 0;
            for (; i <= dim - 4; i += 4) {
                const float4 bv = *(const float4 *)(b + i); // rows are 16-byte aligned when dim % 4 == 0 (checked by the caller)
                const float v0 = qs[i] - bv.x, v1 = qs[i + 1] - bv.y, v2 = qs[i + 2] - bv.z, v3 = qs[i + 3] - bv.w;
                s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
            }
            for (; i < dim; ++i) {
                const float v = qs[i] - b[i];
                s += v * v;
            }
            const float d = sqrtf(s);
            if (d < bd) { bd = d; bi = t; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float od = __shfl_xor(bd, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            const bool take = oi >= 0 && (bi < 0 || od < bd || (od == bd && oi < bi));
            bd = take ? od : bd;
            bi = take ? oi : bi;
        }
        if (lane == 0) { best[q] = bi; bestDist[q] = bd; }
    }
}

extern "C" int pais_seed_match(int device, int nq, const float *query_desc, int nt, const float *train_desc, int dim,
                               int32_t *train_of_query, float *dist)
{
    if (nq < 0 || nt < 0 || dim <= 0 || (nq && (!query_desc || !train_of_query || !dist)) || (nt && !train_desc))
        return sfail("pais_seed_match: bad argument");
    if (device < 0) return sfail("pais_seed_match: needs a GPU");
    if (nq == 0) return 0;
    if (nt == 0) {
        for (int q = 0; q < nq; ++q) { train_of_query[q] = -1; dist[q] = FLT_MAX; }
        return 0;
    }
    if (dim % 4 != 0) return sfail("pais_seed_match: descriptor dimension must be a multiple of 4 (cv::SIFT: 128): rows are read 16 bytes at a time");
    SHIP(hipSetDevice(device));
    struct Bufs { // freed on every return path
        float *dq = nullptr, *dt = nullptr, *dd = nullptr, *dd2 = nullptr;
        int32_t *bq = nullptr, *bt = nullptr;
        ~Bufs() { (void)hipFree(dq); (void)hipFree(dt); (void)hipFree(dd); (void)hipFree(dd2); (void)hipFree(bq); (void)hipFree(bt); }
    } b;
    SHIP(hipMalloc(&b.dq, sizeof(float) * (size_t)nq * dim));
    SHIP(hipMalloc(&b.dt, sizeof(float) * (size_t)nt * dim));
    SHIP(hipMalloc(&b.dd, sizeof(float) * (size_t)nq));
    SHIP(hipMalloc(&b.dd2, sizeof(float) * (size_t)nt));
    SHIP(hipMalloc(&b.bq, sizeof(int32_t) * (size_t)nq));
    SHIP(hipMalloc(&b.bt, sizeof(int32_t) * (size_t)nt));
    SHIP(hipMemcpy(b.dq, query_desc, sizeof(float) * (size_t)nq * dim, hipMemcpyHostToDevice));
    SHIP(hipMemcpy(b.dt, train_desc, sizeof(float) * (size_t)nt * dim, hipMemcpyHostToDevice));
    const size_t lds = sizeof(float) * (size_t)dim;
    hipLaunchKernelGGL(k_nearest_descriptor, dim3(nq < 65536 ? nq : 65536), dim3(64), lds, 0, b.dq, nq, b.dt, nt, dim, b.bq, b.dd);
    hipLaunchKernelGGL(k_nearest_descriptor, dim3(nt < 65536 ? nt : 65536), dim3(64), lds, 0, b.dt, nt, b.dq, nq, dim, b.bt, b.dd2);
    SHIP(hipGetLastError());
    std::vector<int32_t> hq((size_t)nq), ht((size_t)nt);
    SHIP(hipMemcpy(hq.data(), b.bq, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost));
    SHIP(hipMemcpy(ht.data(), b.bt, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost));
    SHIP(hipMemcpy(dist, b.dd, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost));
    // crossCheck: q keeps its nearest train descriptor only if that one's nearest query is q
    for (int q = 0; q < nq; ++q) train_of_query[q] = (hq[q] >= 0 && ht[hq[q]] == q) ? hq[q] : -1;
    return 0;
}

extern "C" int pais_seed_nearest_all(int device, int num_cams, const pais_keypoints *kp, int dim, int32_t *nearest)
{
    if (num_cams < 0 || dim <= 0 || (num_cams && (!kp || !nearest))) return sfail("pais_seed_nearest_all: bad argument");
    if (device < 0) return sfail("pais_seed_nearest_all: needs a GPU");
    if (dim % 4 != 0) return sfail("pais_seed_nearest_all: descriptor dimension must be a multiple of 4 (cv::SIFT: 128): rows are read 16 bytes at a time");
    for (int c = 0; c < num_cams; ++c)
        if (kp[c].n < 0 || (kp[c].n && !kp[c].desc)) return sfail("pais_seed_nearest_all: bad keypoints");
    SHIP(hipSetDevice(device));
    struct Bufs { // freed on every return path
        std::vector<float *> desc;
        int32_t *best = nullptr;
        float *dist = nullptr;
        ~Bufs() { for (float *p : desc) (void)hipFree(p); (void)hipFree(best); (void)hipFree(dist); }
    } b;
    b.desc.assign((size_t)num_cams, nullptr);
    size_t total = 0;
    int nmax = 0;
    for (int c = 0; c < num_cams; ++c) { // every camera's descriptors go up ONCE
        if (kp[c].n == 0) continue;
        SHIP(hipMalloc(&b.desc[c], sizeof(float) * (size_t)kp[c].n * dim));
        SHIP(hipMemcpy(b.desc[c], kp[c].desc, sizeof(float) * (size_t)kp[c].n * dim, hipMemcpyHostToDevice));
        total += (size_t)kp[c].n * (size_t)(num_cams - 1);
        nmax = kp[c].n > nmax ? kp[c].n : nmax;
    }
    if (total == 0) return 0;
    SHIP(hipMalloc(&b.best, sizeof(int32_t) * total));
    SHIP(hipMalloc(&b.dist, sizeof(float) * (size_t)nmax));
    const size_t lds = sizeof(float) * (size_t)dim;
    size_t off = 0;
    for (int i = 0; i < num_cams; ++i)
        for (int j = 0; j < num_cams; ++j) { // nearest(i -> j) once per ORDERED pair
            if (i == j || kp[i].n == 0) continue;
            if (kp[j].n == 0) SHIP(hipMemsetAsync(b.best + off, 0xff, sizeof(int32_t) * (size_t)kp[i].n, 0)); // -1: nothing to match
            else hipLaunchKernelGGL(k_nearest_descriptor, dim3(kp[i].n < 65536 ? kp[i].n : 65536), dim3(64), lds, 0, b.desc[i], kp[i].n,
                                    b.desc[j], kp[j].n, dim, b.best + off, b.dist);
            off += (size_t)kp[i].n;
        }
    SHIP(hipGetLastError());
    SHIP(hipMemcpy(nearest, b.best, sizeof(int32_t) * total, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int pais_seed_fundamental(const pais_camera_desc *from, const pais_camera_desc *to, double F[9])
{
    if (!from || !to || !F) return sfail("pais_seed_fundamental: bad argument");
    // P = [KR | KT] (camera.cpp:123-127)
    double PF[3][4], PT[3][4];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            PF[r][c] = from->KR[r * 3 + c];
            PT[r][c] = to->KR[r * 3 + c];
        }
        PF[r][3] = from->KT[r];
        PT[r][3] = to->KT[r];
    }
    const double cF[4] = {from->center[0], from->center[1], from->center[2], 1.0};
    double eT[3];
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += PT[r][k] * cF[k];
        eT[r] = acc;
    }
    const double ex[3][3] = {{0, -eT[2], eT[1]}, {eT[2], 0, -eT[0]}, {-eT[1], eT[0], 0}};
    // pinv(PF): column k of pinv(PF^T) solves PF^T x = e_k in the least-squares sense (the SVD back-substitution of
    // Mat::inv(DECOMP_SVD)); pinv(PF) is its transpose
    double pinv[4][3];
    for (int k = 0; k < 4; ++k) {
        double At[4][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) At[c][r] = PF[r][c];
        double e[4] = {0, 0, 0, 0}, x[3];
        e[k] = 1.0;
        pais::jacobi_lstsq<4, 3>(At, e, x);
        for (int r = 0; r < 3; ++r) pinv[k][r] = x[r];
    }
    double M[3][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += ex[r][k] * PT[k][c];
            M[r][c] = acc;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += M[r][k] * pinv[k][c];
            F[r * 3 + c] = acc;
        }
    return 0;
}
