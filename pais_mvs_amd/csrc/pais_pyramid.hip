// pais_pyramid.hip -- camera pyramid + edge maps on the GPU (include/pais_pyramid.h; camera.cpp:45-136).
//
// Byte / double streaming work, HBM bound: every kernel is a coalesced row-major sweep (consecutive lanes =
// consecutive pixels of a row).  Per level the source is level 0 (the reference resizes level 0 every time), so a
// level costs one read of the W x H bytes + dh x W x 8 B of the row-reduced intermediate written and read once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/pais_pyramid.h"

namespace {
thread_local std::string g_err;
int pfail(const char *m)
{
    g_err = m;
    return -1;
}
#define PCHK(x)                                                                  \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); rc = -2; goto done; } \
    } while (0)

// area-interpolation weights of one axis for a scale factor fx < 1 (OpenCV's area-resize decimation table for a
// fractional scale; same statements as camera.py:_area_matrix): per destination index a first source index, a tap
// count and `taps` weights normalised to sum 1
struct AxisTable {
    int dsize = 0, maxTaps = 0;
    std::vector<int> first, count;
    std::vector<double> w; // dsize x maxTaps
};
AxisTable area_table(int ssize, double fx)
{
    AxisTable t;
    int dsize = (int)lrint((double)ssize * fx); // saturate_cast<int> == cvRound
    if (dsize < 1) dsize = 1;
    const double scale = 1.0 / fx;
    std::vector<std::vector<double>> vals((size_t)dsize);
    t.first.assign((size_t)dsize, 0);
    t.count.assign((size_t)dsize, 0);
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        std::vector<double> &v = vals[(size_t)dx];
        int first = sx1;
        if (sx1 - fsx1 > 1e-3) {
            first = sx1 - 1;
            v.push_back((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; ++sx) v.push_back(1.0 / cell);
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.0) a = 1.0;
            if (a > cell) a = cell;
            v.push_back(a / cell);
        }
        double rs = 0;
        for (double x : v) rs += x;
        if (rs == 0) rs = 1.0;
        const double inv = 1.0 / rs; // rows normalised exactly like diags(1/rs) @ W
        for (double &x : v) x = inv * x;
        t.first[(size_t)dx] = first;
        t.count[(size_t)dx] = (int)v.size();
        if ((int)v.size() > t.maxTaps) t.maxTaps = (int)v.size();
    }
    t.dsize = dsize;
    t.w.assign((size_t)dsize * t.maxTaps, 0.0);
    for (int dx = 0; dx < dsize; ++dx)
        for (size_t k = 0; k < vals[(size_t)dx].size(); ++k) t.w[(size_t)dx * t.maxTaps + k] = vals[(size_t)dx][k];
    return t;
}

// rows: tmp[dy][x] = sum_k wy[dy][k] * img[first[dy] + k][x]   (ascending k)
__global__ __launch_bounds__(256) void k_area_rows(const uint8_t *img, int w, int h, const int *first, const int *count,
                                                   const double *wy, int maxTaps, int dh, double *tmp)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y;
    if (x >= w || dy >= dh) return;
    const int f = first[dy], c = count[dy];
    double s = 0;
    for (int k = 0; k < c; ++k) s += wy[(size_t)dy * maxTaps + k] * (double)img[(size_t)(f + k) * w + x];
    tmp[(size_t)dy * w + x] = s;
}
// columns + round-half-even + clip: out[dy][dx] = rint(sum_j wx[dx][j] * tmp[dy][first[dx] + j])
__global__ __launch_bounds__(256) void k_area_cols(const double *tmp, int w, const int *first, const int *count, const double *wx,
                                                   int maxTaps, int dw, int dh, uint8_t *out)
{
    const int dx = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw || dy >= dh) return;
    const int f = first[dx], c = count[dx];
    double s = 0;
    for (int j = 0; j < c; ++j) s += wx[(size_t)dx * maxTaps + j] * tmp[(size_t)dy * w + f + j];
    double r = rint(s);
    r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
    out[(size_t)dy * dw + dx] = (uint8_t)r;
}
// Sobel(ksize = 1): central differences with reflect-101 borders, magnitude; min / max of the level through
// ordered-bit atomics (magnitudes are >= 0, so their bit patterns order like the values).  A workgroup sweeps a
// 256 x SOBEL_ROWS tile and issues ONE atomic pair (all waves hammering two addresses was 90 % of the build time).
#define SOBEL_ROWS 16
__global__ __launch_bounds__(256) void k_sobel_mag(const uint8_t *img, int w, int h, double *mag, unsigned long long *minmax)
{
    __shared__ unsigned long long red[8];
    const int x = blockIdx.x * 256 + threadIdx.x;
    unsigned long long lo = ~0ULL, hi = 0ULL;
    if (x < w) {
        const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
        for (int r = 0; r < SOBEL_ROWS; ++r) {
            const int y = blockIdx.y * SOBEL_ROWS + r;
            if (y >= h) break;
            const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
            const double gx = (double)img[(size_t)y * w + xr] - (double)img[(size_t)y * w + xl];
            const double gy = (double)img[(size_t)yd * w + x] - (double)img[(size_t)yu * w + x];
            const double m = sqrt(gx * gx + gy * gy);
            mag[(size_t)y * w + x] = m;
            const unsigned long long u = (unsigned long long)__double_as_longlong(m);
            lo = u < lo ? u : lo;
            hi = u > hi ? u : hi;
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long ol = __shfl_xor(lo, s, 64), oh = __shfl_xor(hi, s, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave] = lo;
        red[4 + wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) {
            lo = red[q] < lo ? red[q] : lo;
            hi = red[4 + q] > hi ? red[4 + q] : hi;
        }
        if (lo != ~0ULL) atomicMin(&minmax[0], lo);
        atomicMax(&minmax[1], hi);
    }
}
__global__ __launch_bounds__(256) void k_edge_normalise(double *mag, size_t n, const unsigned long long *minmax)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double mn = __longlong_as_double((long long)minmax[0]), mx = __longlong_as_double((long long)minmax[1]);
    mag[i] = (mx > mn) ? (mag[i] - mn) / (mx - mn) : 0.0;
}
} // namespace

extern "C" const char *pais_pyramid_last_error(void) { return g_err.c_str(); }

extern "C" void pais_pyramid_free(pais_pyramid *p)
{
    if (!p) return;
    for (int l = 0; l < PAIS_PYRAMID_MAX_LEVELS; ++l) {
        free(p->image[l]);
        free(p->edge[l]);
    }
    free(p);
}

extern "C" int pais_pyramid_build(int device, const uint8_t *level0, int width, int height, int64_t stride, double lod_ratio,
                                  int cfg_max_lod, int build_edges, pais_pyramid **out)
{
    if (!level0 || !out || width <= 0 || height <= 0 || !(lod_ratio > 0.0 && lod_ratio < 1.0)) return pfail("pais_pyramid_build: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return pfail("pais_pyramid_build: no HIP device (the pyramid construction has no host path)");
    if (stride <= 0) stride = width;
    int rc = 0;
    pais_pyramid *P = (pais_pyramid *)calloc(1, sizeof(pais_pyramid));
    uint8_t *d_img0 = nullptr, *d_lvl = nullptr;
    double *d_tmp = nullptr, *d_mag = nullptr, *d_wy = nullptr, *d_wx = nullptr;
    int *d_fy = nullptr, *d_cy = nullptr, *d_fx = nullptr, *d_cx = nullptr;
    unsigned long long *d_mm = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    float total_ms = 0;
    {
        // camera.cpp:63-64
        const int m = (int)(log((double)(width > height ? width : height)) / log(1.0 / lod_ratio));
        int maxLod = m < cfg_max_lod ? m : cfg_max_lod;
        if (maxLod > PAIS_PYRAMID_MAX_LEVELS - 1) maxLod = PAIS_PYRAMID_MAX_LEVELS - 1;
        if (maxLod < 0) maxLod = 0;
        P->max_lod = maxLod;
    }
    PCHK(hipSetDevice(device));
    PCHK(hipStreamCreate(&st));
    PCHK(hipEventCreate(&e0));
    PCHK(hipEventCreate(&e1));
    {
        const size_t n0 = (size_t)width * height;
        PCHK(hipMalloc(&d_img0, n0));
        PCHK(hipMemcpy2DAsync(d_img0, (size_t)width, level0, (size_t)stride, (size_t)width, (size_t)height, hipMemcpyHostToDevice, st));
        PCHK(hipMalloc(&d_lvl, n0));
        PCHK(hipMalloc(&d_tmp, n0 * sizeof(double)));
        PCHK(hipMalloc(&d_mm, 2 * sizeof(unsigned long long)));
        if (build_edges) PCHK(hipMalloc(&d_mag, n0 * sizeof(double)));
        P->width[0] = width;
        P->height[0] = height;
        P->image[0] = (uint8_t *)malloc(n0);
        for (int y = 0; y < height; ++y) memcpy(P->image[0] + (size_t)y * width, level0 + (size_t)y * stride, (size_t)width);
    }
    for (int l = 0; l <= P->max_lod; ++l) {
        const uint8_t *d_src = d_img0;
        int lw = width, lh = height;
        PCHK(hipEventRecord(e0, st));
        if (l > 0) {
            const double fx = pow(lod_ratio, (double)l);
            const AxisTable ty = area_table(height, fx), tx = area_table(width, fx);
            lw = tx.dsize;
            lh = ty.dsize;
            (void)hipFree(d_wy); (void)hipFree(d_wx); (void)hipFree(d_fy); (void)hipFree(d_cy); (void)hipFree(d_fx); (void)hipFree(d_cx);
            d_wy = d_wx = nullptr; d_fy = d_cy = d_fx = d_cx = nullptr;
            PCHK(hipMalloc(&d_wy, ty.w.size() * sizeof(double)));
            PCHK(hipMalloc(&d_wx, tx.w.size() * sizeof(double)));
            PCHK(hipMalloc(&d_fy, sizeof(int) * (size_t)lh));
            PCHK(hipMalloc(&d_cy, sizeof(int) * (size_t)lh));
            PCHK(hipMalloc(&d_fx, sizeof(int) * (size_t)lw));
            PCHK(hipMalloc(&d_cx, sizeof(int) * (size_t)lw));
            PCHK(hipMemcpyAsync(d_wy, ty.w.data(), ty.w.size() * sizeof(double), hipMemcpyHostToDevice, st));
            PCHK(hipMemcpyAsync(d_wx, tx.w.data(), tx.w.size() * sizeof(double), hipMemcpyHostToDevice, st));
            PCHK(hipMemcpyAsync(d_fy, ty.first.data(), sizeof(int) * (size_t)lh, hipMemcpyHostToDevice, st));
            PCHK(hipMemcpyAsync(d_cy, ty.count.data(), sizeof(int) * (size_t)lh, hipMemcpyHostToDevice, st));
            PCHK(hipMemcpyAsync(d_fx, tx.first.data(), sizeof(int) * (size_t)lw, hipMemcpyHostToDevice, st));
            PCHK(hipMemcpyAsync(d_cx, tx.count.data(), sizeof(int) * (size_t)lw, hipMemcpyHostToDevice, st));
            PCHK(hipStreamSynchronize(st)); // the tables are locals
            PCHK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_area_rows, dim3((width + 255) / 256, lh), dim3(256), 0, st, d_img0, width, height, d_fy, d_cy, d_wy,
                               ty.maxTaps, lh, d_tmp);
            hipLaunchKernelGGL(k_area_cols, dim3((lw + 255) / 256, lh), dim3(256), 0, st, d_tmp, width, d_fx, d_cx, d_wx, tx.maxTaps,
                               lw, lh, d_lvl);
            PCHK(hipGetLastError());
            d_src = d_lvl;
            P->width[l] = lw;
            P->height[l] = lh;
            P->image[l] = (uint8_t *)malloc((size_t)lw * lh);
        }
        if (build_edges) {
            const unsigned long long init[2] = {~0ULL, 0ULL};
            PCHK(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_sobel_mag, dim3((lw + 255) / 256, (lh + SOBEL_ROWS - 1) / SOBEL_ROWS), dim3(256), 0, st, d_src, lw, lh, d_mag, d_mm);
            const size_t n = (size_t)lw * lh;
            hipLaunchKernelGGL(k_edge_normalise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_mag, n, d_mm);
            PCHK(hipGetLastError());
            P->edge[l] = (double *)malloc(n * sizeof(double));
        }
        PCHK(hipEventRecord(e1, st));
        if (l > 0) PCHK(hipMemcpyAsync(P->image[l], d_lvl, (size_t)lw * lh, hipMemcpyDeviceToHost, st));
        if (build_edges) PCHK(hipMemcpyAsync(P->edge[l], d_mag, (size_t)lw * lh * sizeof(double), hipMemcpyDeviceToHost, st));
        PCHK(hipStreamSynchronize(st));
        {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) total_ms += ms;
        }
    }
    P->kernel_ms = total_ms;
done:
    (void)hipFree(d_img0); (void)hipFree(d_lvl); (void)hipFree(d_tmp); (void)hipFree(d_mag); (void)hipFree(d_mm);
    (void)hipFree(d_wy); (void)hipFree(d_wx); (void)hipFree(d_fy); (void)hipFree(d_cy); (void)hipFree(d_fx); (void)hipFree(d_cx);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (rc) {
        pais_pyramid_free(P);
        return rc;
    }
    *out = P;
    return 0;
}
