// pais_tile.hpp -- PAIS::getFitness (TMVS/mvs/patch.cpp:914-1047) for patches seen by MANY cameras, with the image
// footprints of the window staged in LDS.  Device-only; included by pais_kernels.hip after pais_eval.hpp.
//
// Why.  With more than a dozen cameras a cost evaluation is S*S * M bilinear taps into M different images (dome rig,
// BASELINE configs[4]: 2601 x 31 = 80 k taps per evaluation, 4.5 GB of pyramids): every tap of the one-wave-per-evaluation
// kernels (pais_eval.hpp) is a pair of dependent 2-byte gathers that miss L1 / L2, the colour rows of a pixel (M x 512 B of
// LDS per wave) leave 1.75 waves per SIMD to hide that latency, and the waves wait 61 % of their cycles
// (profiles/r02_pmc_dome.txt).  But the particles of one candidate tap the SAME few thousand pixels of every camera: the
// window under all their homographies covers an image region barely larger than the window itself.
//
// Mapping (north_star: "image pyramids staged in LDS, one workgroup per patch, PSO particles evaluated as a batched
// map-reduce per patch").  One workgroup = one candidate x up to TILE_WAVES particles of one PSO iteration, one wave per
// particle.  The window is cut into strips of whole 64-pixel steps; per strip
//   1. every wave maps the strip's bounding rectangle through its particle's homographies (one camera per lane) and
//      merges the image-space bounding boxes per camera with LDS atomics;
//   2. wave 0 lays the per-camera tiles out in the tile area (exclusive scan of their sizes);
//   3. all waves copy the tiles' rows global -> LDS with coalesced dword loads (the only global image traffic);
//   4. every wave walks the strip's steps for its particle: the taps are 2-byte LDS reads at (py - y0) * tw + (px - x0).
// The colours of a pixel stay in REGISTERS (camera pairs statically unrolled; the odd tail's colours in three small LDS
// rows), so the workgroup's LDS is the tiles + 2.5 KB of homographies per wave: 8 waves per CU at <= 256 VGPRs.
// Same arithmetic, same operation order and same reduction shape as eval_window<1, false, true, true>: identical bits
// (tests/test_gpu_parity.py: test_tile_kernel_*).
//
// What does NOT go through the tiles, exactly as before:
//   * a particle whose window corners do not map inside every image with one sign of the denominator (corners_inside):
//     it is flagged "pending" and the k_pso_eval2 launch that follows evaluates it with the checked walk;
//   * a camera whose tile does not fit the tile area any more (huge magnification): tapped from global memory.
#pragma once

#define TILE_WAVES 8
#define TILE_MAX_CAMS 32          // cameras tapped per pixel (M <= 32): colours in registers, 16 pairs (or fewer + a tail of <= 3)
#define TILE_STRIP_STEPS 11       // 64-pixel steps per strip (r = 25: 41 steps -> 4 strips of ~14 window rows)

struct TileCam {                  // one camera's tile of the current strip (LDS, 16 bytes: one ds_read_b128)
    int32_t base;                 // byte index in the tile area of image pixel (0, 0): off - y0 * tw - x0
    int32_t tw;                   // row stride of the tile in bytes (multiple of 4); 0: not staged, tap global memory
    int32_t x0, y0;
};
struct TileBox { int32_t xmin, ymin, xmax, ymax; };

__host__ __device__ inline size_t tile_fixed_lds_bytes(int Kmax)
{
    size_t b = eval_block_bytes(Kmax);                                   // EvalPatch + EvalCam[Kmax], shared by the waves
    b += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE_WAVES;     // homographies, per wave
    b += sizeof(double) * 64 * 3 * TILE_WAVES;                           // colours of the odd tail (<= 3 cameras), per wave
    b += (sizeof(TileCam) + sizeof(TileBox)) * (size_t)Kmax + 64;        // tile table, boxes, flags
    return (b + 15) & ~(size_t)15;
}

// one camera group of one window pixel from the tiles: the statements of tap_group<G, 1, false, true> with LDS rows
template <int G>
__device__ __forceinline__ void tile_tap_group(const DevScene &sc, const EvalCam *cams, const TileCam *tcam, const unsigned char *tiles,
                                               const double *Hbuf, int c0, double x, double y, double *col, double &sum)
{
    asm volatile("" : "+v"(x), "+v"(y));
    double nx[G], ny[G], w[G], rw[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const double2 *H2 = (const double2 *)__builtin_assume_aligned(Hbuf + PAIS_H_STRIDE * (c0 + u), 16);
        const double2 ha = H2[0], hb = H2[1], hc = H2[2], hd = H2[3], he = H2[4];
        w[u] = fma(hd.y, y, fma(hd.x, x, he.x));
        nx[u] = fma(ha.y, y, fma(ha.x, x, hb.x));
        ny[u] = fma(hc.x, y, fma(hb.y, x, hc.y));
    }
    if (G == 3) {
        const double p01 = w[0] * w[G > 1 ? 1 : 0];
        const double r = rcp_cr(p01 * w[G - 1]);
        rw[G - 1] = r * p01;
        const double r01 = r * w[G - 1];
        rw[0] = r01 * w[G > 1 ? 1 : 0];
        rw[G > 1 ? 1 : 0] = r01 * w[0];
    } else if (G == 2) {
        const double r = rcp_cr(w[0] * w[G - 1]);
        rw[0] = r * w[G - 1];
        rw[G - 1] = r * w[0];
    } else {
        rw[0] = rcp_cr(w[0]);
    }
    uint16_t r0[G], r1[G];
    double bx[G], by[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int c = c0 + u;
        TileCam tc;
        __builtin_memcpy(&tc, __builtin_assume_aligned(&tcam[c], 16), sizeof(tc));
        const double ix = nx[u] * rw[u], iy = ny[u] * rw[u];
        const int px = (int)ix, py = (int)iy;
        bx[u] = __builtin_amdgcn_fract(ix);
        by[u] = __builtin_amdgcn_fract(iy);
        const int tw = __builtin_amdgcn_readfirstlane(tc.tw);
        if (tw != 0) { // wave-uniform: the camera's tile is staged
            const uint32_t a = (uint32_t)(tc.base + py * tw + px);
            __builtin_memcpy(&r0[u], tiles + a, 2);
            __builtin_memcpy(&r1[u], tiles + a + (uint32_t)tw, 2);
        } else {
            TapInfo ti;
            __builtin_memcpy(&ti, __builtin_assume_aligned(&cams[c].imgOff, 16), sizeof(ti));
            const unsigned char *lvl = sc.imgBlob + ti.imgOff;
            const uint32_t off = (uint32_t)py * (uint32_t)ti.w + (uint32_t)px;
            r0[u] = load_row_at<uint16_t>(lvl, off);
            r1[u] = load_row_at<uint16_t>(lvl, off + (uint32_t)ti.w);
        }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        col[u] = lerp3(r0[u], r1[u], bx[u], by[u]);
        sum += col[u];
    }
}

// The evaluation launch of many-camera batches.  Grid: candidates x particle groups of TILE_WAVES; workgroup of TILE_WAVES
// waves.  Writes A.fit[i], or flags the particle pending (A.part[i][0] = 1) for the k_pso_eval2<.., PENDING> launch behind it.
__global__ __launch_bounds__(64 * TILE_WAVES, 2) void k_pso_tile(DevScene sc, unsigned char *states, int n, int Nmax, int Kmax,
                                                                const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win,
                                                                int tileBytes, int groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    size_t o = eval_block_bytes(Kmax);
    double *Hbuf = (double *)(smem + o) + (size_t)wave * Kmax * PAIS_H_STRIDE; o += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE_WAVES;
    double *tailc = (double *)(smem + o) + (size_t)wave * 3 * 64 + lane;      o += sizeof(double) * 64 * 3 * TILE_WAVES;
    TileCam *tcam = (TileCam *)(smem + o);                                      o += sizeof(TileCam) * (size_t)Kmax;
    TileBox *box = (TileBox *)(smem + o);                                       o += sizeof(TileBox) * (size_t)Kmax;
    int *flags = (int *)(smem + o);                                             // [0]: some particle of the group walks the tiles
    unsigned char *tiles = smem + tile_fixed_lds_bytes(Kmax);
    const size_t SB = pso_state_bytes(Nmax);
    const int WS = win_stride(sc);
    const int S = sc.cfg.patchSize, S2 = S * S;
    const int nSteps = (S2 + 63) >> 6;
    const int nwMax = (int)(eval_block_bytes(Kmax) / 8);

    for (int task = blockIdx.x; task < n * groups; task += gridDim.x) {
        const int c = task / groups, g = task - c * groups;
        const int i = g * TILE_WAVES + wave; // this wave's particle
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        const int active = hd->active, Nrun = hd->N;
        if (!active || g * TILE_WAVES >= Nrun) continue; // uniform over the workgroup
        const bool have = i < Nrun;
        const int iLoad = have ? i : 0;
        const double theta = A.pos[iLoad][0], phi = A.pos[iLoad][1], depth = A.pos[iLoad][2];
        __syncthreads(); // the previous task's LDS is no longer read
        {
            const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
            uint64_t *dst = (uint64_t *)smem;
            for (int q = threadIdx.x; q < nwMax; q += 64 * TILE_WAVES) dst[q] = src[q];
            if (threadIdx.x == 0) flags[0] = 0;
        }
        __syncthreads();
        const int M = ep->M, K = ep->K;
        const WinPix *wbase = win + (size_t)c * WS;

        // ---- the particle: normal, early exits, homographies (the statements of eval_fitness_parts)
        int state = have ? 0 : 3; // 0: walks the tiles, 1: DBL_MAX, 2: pending (checked walk by k_pso_eval2), 3: no particle
        if (have) {
            double nrm[3];
            wave_spherical2normal(theta, phi, nrm, lane);
            const double on[3] = {ep->optNref[0], ep->optNref[1], ep->optNref[2]};
            if (dot3(nrm, on) > 0 || !ep->valid || !(fabs(depth) > 0)) {
                state = 1;
            } else {
                double center[3];
                for (int q = 0; q < 3; ++q) center[q] = ep->ray[q] * depth + ep->Cref[q];
                const double d = -dot3(center, nrm);
                double Mref[9], invH[9], kr[9], kt[3];
                for (int q = 0; q < 9; ++q) kr[q] = ep->KRref[q];
                for (int q = 0; q < 3; ++q) kt[q] = ep->KTref[q];
                plane_matrix(d, ep->lodScale, kr, kt, nrm, Mref);
                inv3(Mref, invH);
                for (int cc = lane; cc < M; cc += 64) {
                    double H[9];
                    if (cams[cc].cam == ep->refCam) {
                        H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
                    } else {
                        double Mc[9];
                        for (int q = 0; q < 9; ++q) kr[q] = cams[cc].KR[q];
                        for (int q = 0; q < 3; ++q) kt[q] = cams[cc].KT[q];
                        plane_matrix(d, ep->lodScale, kr, kt, nrm, Mc);
                        mul33(Mc, invH, H);
                    }
                    for (int q = 0; q < 9; ++q) Hbuf[cc * PAIS_H_STRIDE + q] = H[q];
                }
                wave_sync();
                if (!corners_inside(ep, cams, Hbuf, S, lane)) state = 2;
            }
            if (lane == 0) {
                if (state == 1) A.fit[i] = DBL_MAX;
                A.part[i][0] = (state == 2) ? 1.0 : 0.0;
                if (state == 0) flags[0] = 1;
            }
        }
        __syncthreads();
        if (!flags[0]) continue; // nobody walks the tiles (uniform)

        const double a0 = ep->a0, b0 = ep->b0;
        const double invDiffW = 1.0 / sc.cfg.diffWeighting;
        const bool useDiff = sc.cfg.adaptiveDifferenceEnable != 0;
        const bool hasRef = ep->hasRef != 0;
        const double invK = 1.0 / (double)K;
        double accF[4] = {0, 0, 0, 0}, accW[4] = {0, 0, 0, 0};
        const int nPairs = (M >= 2) ? ((M & 1) ? (M - 3) / 2 : M / 2) : 0; // cameras 0 .. 2 nPairs - 1 in pairs, then a tail of 3, 1 or 0
        const int tail0 = 2 * nPairs, nTail = M - tail0;

        for (int s0 = 0; s0 < nSteps; s0 += TILE_STRIP_STEPS) {
            const int s1 = min(s0 + TILE_STRIP_STEPS, nSteps);
            // ---- 1. bounding boxes of the strip's rectangle (full window rows ya .. yb) in every camera
            for (int q = threadIdx.x; q < M; q += 64 * TILE_WAVES) box[q] = TileBox{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
            __syncthreads();
            if (state == 0) {
                const int ya = (64 * s0) / S, yb = (min(64 * s1, S2) - 1) / S;
                for (int cc = lane; cc < M; cc += 64) {
                    const double *H = Hbuf + PAIS_H_STRIDE * cc;
                    int xmin = INT_MAX, ymin = INT_MAX, xmax = INT_MIN, ymax = INT_MIN;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double x = a0 + (double)((k & 1) ? (S - 1) : 0), y = b0 + (double)((k & 2) ? yb : ya);
                        const double rw = rcp_cr(fma(H[7], y, fma(H[6], x, H[8])));
                        const int qx = (int)(fma(H[1], y, fma(H[0], x, H[2])) * rw), qy = (int)(fma(H[4], y, fma(H[3], x, H[5])) * rw);
                        xmin = min(xmin, qx); xmax = max(xmax, qx);
                        ymin = min(ymin, qy); ymax = max(ymax, qy);
                    }
                    atomicMin(&box[cc].xmin, xmin); atomicMin(&box[cc].ymin, ymin);
                    atomicMax(&box[cc].xmax, xmax); atomicMax(&box[cc].ymax, ymax);
                }
            }
            __syncthreads();
            // ---- 2. layout: one pixel of margin against the rounding of the corner quotients, +1 for the bilinear neighbour
            if (wave == 0) {
                int carry = 0;
                for (int cb = 0; cb < M; cb += 64) {
                    const int cc = cb + lane;
                    int x0 = 0, y0 = 0, tw = 0, th = 0;
                    if (cc < M && box[cc].xmax >= box[cc].xmin) {
                        const int lw = cams[cc].w, lh = cams[cc].h;
                        x0 = max(box[cc].xmin - 1, 0); y0 = max(box[cc].ymin - 1, 0);
                        const int x1 = min(box[cc].xmax + 2, lw - 1), y1 = min(box[cc].ymax + 2, lh - 1);
                        tw = ((x1 - x0 + 1) + 3) & ~3;
                        th = y1 - y0 + 1;
                    }
                    int sz = tw * th, incl = sz;
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {
                        const int up = __shfl_up(incl, m, 64);
                        incl += (lane >= m) ? up : 0;
                    }
                    const int off = carry + incl - sz;
                    const bool fits = sz > 0 && off + sz <= tileBytes;
                    if (cc < M) {
                        TileCam tc;
                        tc.tw = fits ? tw : 0;
                        tc.base = fits ? (off - y0 * tw - x0) : 0;
                        tc.x0 = fits ? x0 : 0;
                        tc.y0 = fits ? (y0 | (th << 16)) : 0; // rows of the tile in the upper half (levels are < 65536 high)
                        tcam[cc] = tc;
                    }
                    carry += __shfl(incl, 63, 64);
                }
            }
            __syncthreads();
            // ---- 3. copy: rows dealt to half waves, dwords of a row to their lanes
            for (int cc = 0; cc < M; ++cc) {
                const TileCam tc = tcam[cc];
                if (tc.tw == 0) continue;
                const int th = tc.y0 >> 16, y0 = tc.y0 & 0xffff, twd = tc.tw >> 2;
                const unsigned char *lvl = sc.imgBlob + cams[cc].imgOff;
                const uint32_t lw = (uint32_t)cams[cc].w;
                unsigned char *dst = tiles + (tc.base + y0 * tc.tw + tc.x0);
                for (int r = 2 * wave + (lane >> 5); r < th; r += 2 * TILE_WAVES) {
                    const unsigned char *srow = lvl + (size_t)(uint32_t)(y0 + r) * lw + (uint32_t)tc.x0;
                    for (int dw = lane & 31; dw < twd; dw += 32) {
                        uint32_t v;
                        __builtin_memcpy(&v, srow + 4 * dw, 4);
                        *(uint32_t *)(dst + r * tc.tw + 4 * dw) = v;
                    }
                }
            }
            __syncthreads();
            // ---- 4. the strip's steps for this wave's particle
            if (state == 0) {
                int kpix = 64 * s0 + lane;
                int yw = kpix / S, xw = kpix - yw * S;
                const int qA = 64 / S, rA = 64 - qA * S;
                for (int st = s0; st < s1; ++st) {
                    const WinPix wp = wbase[64 * st + lane]; // (the padding lanes of the last step are masked entries)
                    const double x = a0 + (double)xw, y = b0 + (double)yw;
                    xw += rA; yw += qA;
                    yw += (xw >= S) ? 1 : 0;
                    xw -= (xw >= S) ? S : 0;
                    double sum = hasRef ? wp.refCol : 0.0;
                    double col[TILE_MAX_CAMS];
#pragma unroll
                    for (int u = 0; u < TILE_MAX_CAMS / 2; ++u) {
                        if (u < nPairs) tile_tap_group<2>(sc, cams, tcam, tiles, Hbuf, 2 * u, x, y, &col[2 * u], sum);
                        else { col[2 * u] = 0; col[2 * u + 1] = 0; }
                    }
                    if (nTail == 3) {
                        double t3[3];
                        tile_tap_group<3>(sc, cams, tcam, tiles, Hbuf, tail0, x, y, t3, sum);
                        tailc[0] = t3[0]; tailc[64] = t3[1]; tailc[128] = t3[2];
                    } else if (nTail == 1) {
                        double t1[1];
                        tile_tap_group<1>(sc, cams, tcam, tiles, Hbuf, tail0, x, y, t1, sum);
                        tailc[0] = t1[0];
                    }
                    const double mean = sum * invK;
                    double sad = hasRef ? fabs(wp.refCol - mean) : 0.0;
#pragma unroll
                    for (int u = 0; u < TILE_MAX_CAMS / 2; ++u) {
                        if (u < nPairs) {
                            sad += fabs(col[2 * u] - mean);
                            sad += fabs(col[2 * u + 1] - mean);
                        }
                    }
                    for (int q = 0; q < nTail; ++q) sad += fabs(tailc[64 * q] - mean);
                    const bool act = wp.wStat >= 0.0;
                    const double sadq = sad * invK;
                    double weight = wp.wStat;
                    if (useDiff) weight *= det_exp_poly(-(sadq * sadq) * invDiffW);
                    const int ga = st & 3; // canonical sub-accumulator of the step (uniform)
#define PAIS_TACC(a)                                          \
    {                                                         \
        accW[a] = act ? (accW[a] + weight) : accW[a];         \
        accF[a] = act ? fma(weight, sadq, accF[a]) : accF[a]; \
    }
                    if (ga == 0) PAIS_TACC(0) else if (ga == 1) PAIS_TACC(1) else if (ga == 2) PAIS_TACC(2) else PAIS_TACC(3)
#undef PAIS_TACC
                }
            }
            // (the barrier at the head of the next strip / task orders these reads before the tiles are overwritten)
        }
        if (state == 0) {
            double f4[4], w4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f4[a] = wave_sum(accF[a]);
                w4[a] = wave_sum(accW[a]);
            }
            if (lane == 0) A.fit[i] = combine_parts(f4, w4);
        }
    }
}
