// pais_eval.hpp -- PAIS::getFitness (TMVS/mvs/patch.cpp:914-1047) for one particle, executed by one wave64.
// Device-only; included by pais_kernels.hip.
//
// What a PSO run shares ("evaluation block", built ONCE per run by build_eval_block):
//   * EvalPatch / EvalCam[M]: the matrices of the reference camera and of the M visible cameras other than the
//     reference camera, the image offsets at the patch's LOD;
//   * the reference window.  Every particle's centre is ray * depth + C_ref (patch.cpp:944): it lies on the reference
//     camera's ray, so its projection into the reference camera (patch.cpp:952) is the same point for every particle
//     and iteration of the run up to rounding.  The window origin (a0, b0) = project(ray + C_ref) - r is therefore
//     evaluated once, and with it everything the cost reads from the reference camera per window pixel: the mask
//     (patch.cpp:986), the reference camera's own colour (its homography is the identity, patch.cpp:317-320: a plain
//     bilinear sample at (x, y)), the distance weight (patch.cpp:1031) and the gradient weight (patch.cpp:1037).
//     They are stored as WinPix{refCol, wStat}[S*S] (wStat = -1 for a masked pixel) and read back coalesced.
// Per particle: the plane-induced homographies of the M other cameras (patch.cpp:290-330), S*S * M bilinear taps,
// mean / mean absolute deviation over the K colours, the difference weight, the weighted sums.
//
// "Kernel arithmetic" v5 (the CPU checker mirrors it statement by statement in its kernel-arithmetic
// mode): homography rows with fma; ONE reciprocal per camera group (pairs, one triple for an odd count); tap bounds
// tested on the truncated integer coordinate; bilinear as three lerps a + f (b - a); colours summed reference first,
// then the other cameras in camIdx order (from 13 cameras on in two groups: PAIS_TWO_LEVEL_K below); mean and SAD scaled by 1/K; weight = wStat * exp_poly(-sad^2 / diffW);
// lane partial sums into four canonical sub-accumulators (64-pixel step mod 4), wave64 xor butterfly,
// ((a0 + a1) + a2) + a3.
#pragma once

// ------------------------------------------------------- evaluation block ----
struct EvalCam {        // one visible camera other than (the first occurrence of) the reference camera
    double KR[9];
    double KT[3];
    // what a tap reads, one aligned 16-byte LDS read (ds_read_b128: wave-uniform operands cost LDS cycles, not VALU)
    uint64_t imgOff;    // into DevScene::imgBlob / imgF at the patch's LOD
    int w;
    uint32_t qpack;     // (w - 4) | (h - 4) << 16: largest truncated tap coordinates that pass patch.cpp:999
    int h, cam, pad0, pad1;
};
struct TapInfo { uint64_t imgOff; int w; uint32_t qpack; };
static_assert(sizeof(EvalCam) == 128 && offsetof(EvalCam, imgOff) == 96, "EvalCam layout");
#define PAIS_H_STRIDE 10 // doubles per homography in LDS: 9 + 1 padding, so that rows are 16-byte aligned (ds_read_b128)
// Kernel arithmetic, round 6: a patch seen by PAIS_TWO_LEVEL_K or more cameras sums its colours -- and their absolute deviations
// from the mean -- in TWO groups: the reference colour and the first h = 2 * ((M + 4) / 4) of the M other cameras, then the
// remaining ones (the second group is about two cameras smaller: its owner in the split tile kernel also finishes the pixel);
// each group sequentially in camIdx order, the two group sums added once.  (Fewer cameras: one sequential sum,
// as before -- every golden vector of the pawn and ring scenes is unchanged.)  Why: two waves can then each own a group of a
// particle's cameras and exchange two partial sums per pixel step instead of handing a running sum back and forth three times
// (pais_tile2.hpp; the sequential form cost the split kernel everything its occupancy bought: profiles/r06_tile2_diag.txt).
#define PAIS_TWO_LEVEL_K 13
__host__ __device__ inline int two_level_split(int K, int M) { return K >= PAIS_TWO_LEVEL_K ? 2 * ((M + 4) / 4) : M; }
struct EvalPatch {
    double ray[3], Cref[3], optNref[3], KRref[9], KTref[3];
    double lodScale;
    double a0, b0;      // window origin of the run: project(ray + C_ref) - patchRadius
    int M;              // cameras in EvalCam[]
    int K;              // visible cameras of the patch (divisor of mean / SAD)
    int hasRef;         // the reference camera is among the visible ones (always, for patches refine() builds)
    int valid;          // the window lies inside [2, dim-3) of the reference level (patch.cpp:952-962)
    int LOD, refCam;
    int refPos;         // position in camIdx of the first occurrence of the reference camera (-1: none) -- the literal arithmetic
                        // (pais_literal.hpp) adds the colours in camIdx order
    int pad16;          // sizeof % 16 == 0: EvalCam[] and the homographies behind it stay 16-byte aligned
};
static_assert(sizeof(EvalPatch) % 16 == 0, "EvalPatch layout");
struct WinPix {
    double refCol;      // bilinear sample of the reference level at (x, y)
    double wStat;       // [dist] G(x, y) * [grad] exp(-1 / (edge * gradientWeighting)); -1: masked pixel / padding
};
// WinPix entries per candidate: S*S rounded up to whole 64-pixel steps; the padding entries are "masked", so that the
// evaluation needs no lane-level validity test
// v * u with u as the instruction's scalar operand (u wave-uniform): keeps u out of the vector registers for good -- left to
// the compiler, a uniform double that feeds a VALU instruction is copied to a VGPR pair and kept (or spilled) there
__device__ __forceinline__ double mul_uniform(double v, double u)
{
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(v), "s"(u));
    return r;
}
// a double that every lane holds alike, moved to an SGPR pair (the compiler cannot prove it uniform when it comes from LDS or a
// VALU division): SGPR operands cost no vector register
__device__ __forceinline__ double uniform_d(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__host__ __device__ inline int win_stride(const DevScene &sc) { return (sc.cfg.patchSize * sc.cfg.patchSize + 63) & ~63; }

// median of three == clamp(v, lo, hi) for lo <= hi, one instruction
__device__ __forceinline__ int clamp_i32(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// one row of a bilinear tap: the left pixel and the (exact) difference to its right neighbour, with one global load.
// BYTES = false: from the float2 copy {I, dI}; BYTES = true: two adjacent bytes of the byte blob (pais_internal.h).
template <bool BYTES> struct Tap;
template <> struct Tap<false> {
    typedef PaisImgT Row;
    static constexpr uint32_t kElem = sizeof(PaisImgT);
    static __device__ __forceinline__ const unsigned char *blob(const DevScene &sc) { return (const unsigned char *)sc.imgF; }
};
template <> struct Tap<true> {
    typedef uint16_t Row;
    static constexpr uint32_t kElem = 1;
    static __device__ __forceinline__ const unsigned char *blob(const DevScene &sc) { return sc.imgBlob; }
};
// from a wave-uniform level base + a 32-bit BYTE offset inside the level (a level is far below 4 GB): the form
// global_load ... v_off, s[base:base+1] needs no 64-bit VALU address arithmetic
template <class Row> __device__ __forceinline__ Row load_row_at(const unsigned char *levelBase, uint32_t byteOff)
{
    Row v;
    __builtin_memcpy(&v, levelBase + byteOff, sizeof(Row));
    return v;
}

// bilinear as three lerps a + f (b - a) from two rows; the pixel differences are exact (small integers)
__device__ __forceinline__ double lerp3(double i00, double d0, double i01, double d1, double bx, double by)
{
    const double t0 = fma(bx, d0, i00);
    const double t1 = fma(bx, d1, i01);
    return fma(by, t1 - t0, t0);
}
__device__ __forceinline__ double lerp3(float2 r0, float2 r1, double bx, double by)
{
    return lerp3((double)r0.x, (double)r0.y, (double)r1.x, (double)r1.y, bx, by);
}
__device__ __forceinline__ double lerp3(double2 r0, double2 r1, double bx, double by)
{
    return lerp3(r0.x, r0.y, r1.x, r1.y, bx, by);
}
__device__ __forceinline__ double lerp3(uint16_t r0, uint16_t r1, double bx, double by)
{
    const int a0 = (int)(r0 & 0xff), b0 = (int)(r0 >> 8);
    const int a1 = (int)(r1 & 0xff), b1 = (int)(r1 >> 8);
    return lerp3((double)a0, (double)(b0 - a0), (double)a1, (double)(b1 - a1), bx, by);
}

// 1 / w, correctly rounded for every w whose reciprocal is a normal number: hardware estimate, two Newton steps
// (relative error ~2^-90 before rounding), one Markstein correction q = y + y (1 - w y) -- the sequence the
// compiler's IEEE division ends with, without its range scaling (v_div_scale / v_div_fmas / v_div_fixup: 4 fewer
// instructions per camera group).  Seed independent, i.e. the bits of the CPU's 1.0 / w.  w = 0, inf, NaN, denormal
// give inf / 0 / NaN, never a finite wrong value of ordinary size: the tap is rejected by its bounds either way.
__device__ __forceinline__ double rcp_cr(double w)
{
#if PAIS_RCP_NEWTON
    double y = __builtin_amdgcn_rcp(w);
    double e = fma(-w, y, 1.0);
    y = fma(y, e, y);
    e = fma(-w, y, 1.0);
    y = fma(y, e, y);
    e = fma(-w, y, 1.0);
    return fma(y, e, y);
#else
    return 1.0 / w;
#endif
}

// Camera::pyramidEdge(LOD) at one pixel without a stored map (camera.cpp:72-77, 87-91): Sobel(ksize = 1) = central
// differences with reflect-101 borders, magnitude sqrt(gx^2 + gy^2) (exact: small integers), normalised with the level's
// minimum / maximum -- the statements of k_sobel_mag / k_edge_normalise (pais_pyramid.hip), which reproduce the host
// construction bit for bit.  Saves a double-precision copy of every pyramid (36 GB at 128 x 12.6 MP).
__device__ __forceinline__ double edge_on_the_fly(const uint8_t *img, int w, int h, int x, int y, double mn, double mx)
{
    const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const double gx = (double)img[(size_t)y * w + xr] - (double)img[(size_t)y * w + xl];
    const double gy = (double)img[(size_t)yd * w + x] - (double)img[(size_t)yu * w + x];
    const double m = sqrt(gx * gx + gy * gy);
    return (mx > mn) ? (m - mn) / (mx - mn) : 0.0;
}

// Builds the evaluation block of one PSO run; called by all 64 lanes of one wave.
//   ep, cams : K <= PAIS_MAX_VIS = 64 cameras, one per lane
//   win      : S*S WinPix
__device__ void build_eval_block(const DevScene &sc, EvalPatch *ep, EvalCam *cams, WinPix *win, const double *ray, int refCam,
                                 int LOD, int K, const int *camIdx, int lane)
{
    const DevCamera &rc = sc.cams[refCam];
    // the first occurrence of the reference camera is served from the window block; every other camera gets a slot
    const int myCam = lane < K ? camIdx[lane] : -1;
    const unsigned long long isRef = __ballot(lane < K && myCam == refCam);
    const int firstRef = isRef ? (__ffsll((long long)isRef) - 1) : -1;
    const bool other = lane < K && lane != firstRef;
    const unsigned long long om = __ballot(other);
    const int slot = __popcll(om & ((1ull << lane) - 1ull));
    if (other) {
        const DevCamera &dc = sc.cams[myCam];
        EvalCam &e = cams[slot];
        for (int i = 0; i < 9; ++i) e.KR[i] = dc.KR[i];
        for (int i = 0; i < 3; ++i) e.KT[i] = dc.KT[i];
        e.imgOff = dc.imgOff[LOD];
        e.w = dc.w[LOD];
        e.h = dc.h[LOD];
        e.cam = myCam;
        e.qpack = (uint32_t)((dc.w[LOD] - 4) & 0xffff) | ((uint32_t)((dc.h[LOD] - 4) & 0xffff) << 16); // levels are < 65540 wide
        e.pad0 = e.pad1 = 0;
    }
    const double s = sc.lodScale[LOD];
    const int refW = rc.w[LOD], refH = rc.h[LOD];
    const int r = sc.cfg.patchRadius;
    double pt[2];
    {
        const double c1[3] = {ray[0] * 1.0 + rc.C[0], ray[1] * 1.0 + rc.C[1], ray[2] * 1.0 + rc.C[2]};
        project_raw(rc.R, rc.T, rc.focal, rc.pp, s, c1, pt);
    }
    bool valid = LOD <= rc.maxLOD && in_image_d(pt, refW, refH);                                                  // :952
    if (valid && (pt[0] - r < 2 || pt[0] + r >= refW - 3 || pt[1] - r < 2 || pt[1] + r >= refH - 3)) valid = false; // :957
    const double a0 = pt[0] - r, b0 = pt[1] - r;
    if (lane == 0) {
        for (int i = 0; i < 3; ++i) {
            ep->ray[i] = ray[i];
            ep->Cref[i] = rc.C[i];
            ep->optNref[i] = rc.optN[i];
            ep->KTref[i] = rc.KT[i];
        }
        for (int i = 0; i < 9; ++i) ep->KRref[i] = rc.KR[i];
        ep->lodScale = s;
        ep->a0 = a0;
        ep->b0 = b0;
        ep->M = __popcll(om);
        ep->K = K;
        ep->hasRef = firstRef >= 0 ? 1 : 0;
        ep->refPos = firstRef;
        ep->pad16 = 0;
        ep->valid = valid ? 1 : 0;
        ep->LOD = LOD;
        ep->refCam = refCam;
    }
    if (!valid) return; // no pixel of the window is ever read
    const int S = sc.cfg.patchSize, S2 = S * S;
    if (lane >= (S2 & 63) && (S2 & 63) != 0) { // padding of the last step
        WinPix pad;
        pad.refCol = 0.0;
        pad.wStat = -1.0;
        win[(S2 & ~63) + lane] = pad;
    }
    const uint8_t *refImg = sc.imgBlob + rc.imgOff[LOD];
    const double *refEdge = sc.edgeBlob ? sc.edgeBlob + rc.edgeOff[LOD] : nullptr;
    const double eMin = rc.edgeMin[LOD], eMax = rc.edgeMax[LOD];
    const bool useDist = sc.cfg.adaptiveDistanceEnable != 0, useGrad = sc.cfg.adaptiveGradientEnable != 0;
    const double gradW = sc.cfg.gradientWeighting;
    for (int k = lane; k < S2; k += 64) {
        const int yi = k / S, xi = k - yi * S;
        const double x = a0 + (double)xi, y = b0 + (double)yi; // == the reference's ++x / ++y walk
        const int rx = cv_round(x), ry = cv_round(y);
        const int qx = (int)x, qy = (int)y;                   // 2 <= x < w - 3: truncation == floor
        const double bx = x - (double)qx, by = y - (double)qy;
        const uint32_t off = (uint32_t)qy * (uint32_t)refW + (uint32_t)qx;
        WinPix wp;
        wp.refCol = lerp3(load_row_at<uint16_t>(refImg, off), load_row_at<uint16_t>(refImg, off + (uint32_t)refW), bx, by);
        double ws = useDist ? sc.gauss[xi * S + yi] : 1.0;
        if (useGrad) {
            const double e = refEdge ? refEdge[ry * refW + rx] : edge_on_the_fly(refImg, refW, refH, rx, ry, eMin, eMax);
            ws *= det_exp_poly(-1.0 / (e * gradW));
        }
        wp.wStat = (refImg[ry * refW + rx] != 0) ? ws : -1.0; // :986
        win[k] = wp;
    }
}

// G consecutive cameras of NS window pixels of this lane: homography (fma), ONE reciprocal per (group, pixel),
// bounds test, two 8-byte row loads and three fma lerps per tap.  No lane-dependent branches; the loads of the
// G x NS taps are independent so they overlap.
// CHECK = false: the evaluation has established that no tap of the window can leave the image (corners_inside below):
// no clamping, no flags.
template <int G, int NS, bool CHECK, bool BYTES>
__device__ __forceinline__ void tap_group(const DevScene &sc, const EvalCam *cams, const double *Hbuf, double *myc,
                                          int c0, double *x, double *y, uint32_t *badBits, double *sum)
{
    static_assert(G >= 1 && G <= 3, "kernel arithmetic is defined for groups of 1, 2, 3 cameras");
    // opaque re-definition of the pixel coordinates per group: without it the register allocator splits the live
    // ranges of x and y around the camera loop into one copy per use
#pragma unroll
    for (int q = 0; q < NS; ++q) asm volatile("" : "+v"(x[q]), "+v"(y[q]));
    double bx[NS][G], by[NS][G], nx[NS][G], ny[NS][G], w[NS][G], rw[NS][G];
    typedef typename Tap<BYTES>::Row RowTap;
    constexpr uint32_t kElem = Tap<BYTES>::kElem;
    const unsigned char *base[G];
    uint32_t off[NS][G], cwv[G];
    uint64_t tapWord[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        // five 16-byte LDS reads per homography (half the LDS cycles of nine 8-byte ones)
        const double2 *H2 = (const double2 *)__builtin_assume_aligned(Hbuf + PAIS_H_STRIDE * (c0 + u), 16);
        const double2 ha = H2[0], hb = H2[1], hc = H2[2], hd = H2[3], he = H2[4];
        const double h0 = ha.x, h1 = ha.y, h2 = hb.x, h3 = hb.y, h4 = hc.x, h5 = hc.y, h6 = hd.x, h7 = hd.y, h8 = he.x;
        // the 10th double of the record: where the camera's level starts in the blob (40 bits) and its row length (24 bits),
        // written next to the homography (eval_fitness_parts) -- the taps need no read of the EvalCam record
        tapWord[u] = (uint64_t)__double_as_longlong(he.y);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            w[q][u] = fma(h7, y[q], fma(h6, x[q], h8));
            nx[q][u] = fma(h1, y[q], fma(h0, x[q], h2));
            ny[q][u] = fma(h4, y[q], fma(h3, x[q], h5));
        }
    }
    // one reciprocal per group and pixel (Montgomery batch inversion).  A zero / non-finite w poisons the whole
    // group, which is right: any overflowing tap makes the whole call DBL_MAX.
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        if (G == 3) {
            const double p01 = w[q][0] * w[q][G > 1 ? 1 : 0];
            const double r = rcp_cr(p01 * w[q][G - 1]);
            rw[q][G - 1] = r * p01;                // 1/w2
            const double r01 = r * w[q][G - 1];    // 1/(w0 w1)
            rw[q][0] = r01 * w[q][G > 1 ? 1 : 0];
            rw[q][G > 1 ? 1 : 0] = r01 * w[q][0];
        } else if (G == 2) {
            const double r = rcp_cr(w[q][0] * w[q][G - 1]);
            rw[q][0] = r * w[q][G - 1];
            rw[q][G - 1] = r * w[q][0];
        } else {
            rw[q][0] = rcp_cr(w[q][0]);
        }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int c = c0 + u;
        int qxmax = 0, qymax = 0;
        if (CHECK) { // (the checked walk also reads the tap bounds of the camera)
            const uint32_t qp = cams[c].qpack;
            qxmax = (int)(qp & 0xffffu);
            qymax = (int)(qp >> 16);
        }
        uint32_t cw;
        {   // wave-uniform: keep the base in SGPRs so that the taps are global_load ... v_off, s[base] (no 64-bit VALU address math)
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)tapWord[u]), hi = __builtin_amdgcn_readfirstlane((uint32_t)(tapWord[u] >> 32));
            base[u] = Tap<BYTES>::blob(sc) + (((uint64_t)(hi & 0xffu) << 32) | lo) * kElem;
            cw = hi >> 8;
        }
        cwv[u] = cw;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const double ix = nx[q][u] * rw[q][u], iy = ny[q][u] * rw[q][u];
            // patch.cpp:999 in the integer domain, without branches: for the truncated q = (int)ix,
            // 2 <= ix < w-3  <=>  2 <= q <= w-4 (NaN converts to 0, +-inf / overflow saturate: all rejected;
            // w == 0 needs no test of its own: the reciprocal is inf / NaN).  The clamped q addresses the tap,
            // q != clamp(q) flags the overflow.  frac(ix) == ix - (double)q exactly for an accepted ix.
            const int qx = (int)ix, qy = (int)iy;
            int px = qx, py = qy;
            if (CHECK) {
                px = clamp_i32(qx, 2, qxmax);
                py = clamp_i32(qy, 2, qymax);
                badBits[q] = badBits[q] | (uint32_t)(px ^ qx) | (uint32_t)(py ^ qy);
            }
            bx[q][u] = __builtin_amdgcn_fract(ix);
            by[q][u] = __builtin_amdgcn_fract(iy);
            off[q][u] = (__umul24((uint32_t)py, cw) + (uint32_t)px) * kElem; // byte offset in the level (w < 2^24)
        }
    }
    RowTap r0[NS][G], r1[NS][G];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < G; ++u) {
            r0[q][u] = load_row_at<RowTap>(base[u], off[q][u]);
            r1[q][u] = load_row_at<RowTap>(base[u], off[q][u] + cwv[u] * kElem);
        }
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const double col = lerp3(r0[q][u], r1[q][u], bx[q][u], by[q][u]);
            myc[((c0 + u) * NS + q) * 64] = col;
            sum[q] += col;
        }
}

// LDS scratch of one evaluating wave
//   Hbuf : M*PAIS_H_STRIDE doubles (homographies, patch.cpp:290-330)
//   cbuf : (NS*M + 8)*64 doubles  (per-camera colour of the lane's NS pixels; last 8 rows: the lane's 4 x (fitness,
//          weight) sub-accumulators)
// The lane's four (fitness, weight) sub-accumulators: LDS rows for the two-pixel kernels (few cameras: registers are the
// scarce resource there), registers for the one-pixel kernels (many cameras: the LDS scratch caps the occupancy;
// measured +17 % on the 32-camera ring, -1 % on the 5-camera pawn scene if used there too)
// Three shapes of the evaluation kernels (template parameters NS, ACCR; chosen by the batch's largest camera count):
//   NS 2, LDS accumulators       K <= PAIS_TWO_PIXELS_MAXK (6)   registers are the scarce resource (168 VGPRs at 3 waves / SIMD)
//   NS 2, register accumulators  K <= PAIS_TWO_PIXELS_REG_MAXK (12)  ring (K 7..11): +5.7 % over the one-pixel kernel
//   NS 1, register accumulators  beyond                          the colour rows (NS x M x 512 B) cap the occupancy
// pairs of cameras per trip of the one-pixel kernels' camera loop (dome, K <= 32: 2 with 2 waves / SIMD requested
// +5.8 % patches/s -- the waves wait 60 % of their cycles at the 1.75 waves / SIMD the colour rows leave; 3 and 4 no better)
#ifndef PAIS_WAVE_SUM8
#define PAIS_WAVE_SUM8 1 // 0: eight separate wave_sum_x at the end of an evaluation (round 5; A/B builds)
#endif
#ifndef PAIS_NS1_UNROLL
#define PAIS_NS1_UNROLL 2
#endif
#define PAIS_ACC_IN_REGS(NS) (PAIS_ACC_REG || (NS) == 1)
#define PAIS_CBUF_ROWS(NS, M, ACCR) ((NS) * (M) + ((ACCR) ? 0 : 8))
__host__ __device__ inline size_t eval_block_bytes(int Kmax) { return sizeof(EvalPatch) + sizeof(EvalCam) * (size_t)Kmax; }
__host__ __device__ inline size_t eval_lds_bytes(int NS, int Kmax, bool accr)
{
    return eval_block_bytes(Kmax) + sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax + sizeof(double) * 64 * (size_t)PAIS_CBUF_ROWS(NS, Kmax, accr);
}

// Reduction shape (canonical, independent of how many waves share one evaluation): the 64-pixel steps of the
// window are dealt round-robin to FOUR sub-accumulators (step mod 4); each is summed per lane over its steps,
// butterfly-reduced over the 64 lanes, and the four results are added as ((a0 + a1) + a2) + a3.  One wave
// computes all four (nparts = 1), or `nparts` (2 / 4) waves compute the sub-accumulators a with
// a mod nparts == part and whoever consumes the fitness adds them -- the same bits either way.
// Returns 0 and fills f4/w4 (zeros for the sub-accumulators of other parts), or 1 if the call is DBL_MAX.
template <int NS, bool CHECK, bool BYTES, bool ACCR>
__device__ int eval_window(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *cbuf,
                           const WinPix *win, int lane, int part, int nparts, double *f4, double *w4);

// The taps of a window are the images of its pixels under maps (h0 x + h1 y + h2) / (h6 x + h7 y + h8): where the
// denominator keeps one sign over the window -- it is affine, so: at the four corners -- the image of the (convex) window
// is convex and lies inside the bounding box of the four corner images.  If every camera maps all four corners into
// [3, w-4) x [3, h-4) -- the reference's box shrunk by a pixel, see below -- no tap can leave [2, w-3) x [2, h-3), the whole-call DBL_MAX of patch.cpp:999-1002 cannot trigger and the per-tap
// clamp / flag logic (5 of 39 instructions per tap) is dropped for this evaluation.  Otherwise (a particle that grazes an
// image border, or a degenerate plane) the evaluation runs the checked loop: the reference's rule tap by tap.
// One (corner, camera) pair per lane; the corner taps use a plain quotient n * (1 / w).
__device__ __forceinline__ bool corners_inside(const EvalPatch *ep, const EvalCam *cams, const double *Hbuf, int S, int lane)
{
    const int M = ep->M;
    bool ok = true;
    for (int t0 = 0; t0 < 4 * M; t0 += 64) {
        const int t = t0 + lane;
        const int c = (t < 4 * M) ? (t >> 2) : 0, corner = t & 3;
        const double x = ep->a0 + (double)((corner & 1) ? (S - 1) : 0), y = ep->b0 + (double)((corner & 2) ? (S - 1) : 0);
        const double *H = Hbuf + PAIS_H_STRIDE * c;
        const double w = fma(H[7], y, fma(H[6], x, H[8]));
        const double rw = rcp_cr(w);
        const double ix = fma(H[1], y, fma(H[0], x, H[2])) * rw, iy = fma(H[4], y, fma(H[3], x, H[5])) * rw;
        const int qx = (int)ix, qy = (int)iy;
        const uint32_t qp = cams[c].qpack;
        // one pixel inside the reference's bound [2, w-3) x [2, h-3): the window's taps come out of the batch inversion
        // r * w_other, which rounds differently from this corner's n * (1 / w) -- a corner within an ulp of the bound must not
        // be able to put a tap on the other side of it.  And a denominator of ordinary size: the products of two or three of
        // them that the batch inversion forms can then neither overflow nor underflow.
#if PAIS_CORNER_WTEST
        bool in = qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16) && fabs(w) > 1e-90 && fabs(w) < 1e90;
#else
        bool in = qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16);
#endif
        // one sign of w over the four corners of a camera: lanes 4c .. 4c+3
        const unsigned long long neg = __ballot(w < 0.0), pos = __ballot(w > 0.0);
        const unsigned long long grp = 0xFull << (lane & ~3);
        in = in && (((neg & grp) == 0) || ((pos & grp) == 0)) && (((neg | pos) & grp) == grp);
        ok = ok && (in || t >= 4 * M);
    }
    return __all(ok);
}

// spherical2normal (utility.h:25-29) by one wave: sin(theta), cos(theta), sin(phi), cos(phi) are four evaluations of the same
// fdlibm routine -- shared argument reduction, then k_sin or k_cos by quadrant -- so lanes 0..3 (every lane by its low two
// bits) evaluate one of them each in ONE pass and the four values are broadcast with v_readlane: a quarter of the
// instructions of four wave-uniform calls (a tenth of a cost evaluation at five cameras), the bits of det_sin / det_cos
// (the same functions on the same arguments; cos(x) is sin's quadrant table shifted by one).
__device__ __forceinline__ void wave_spherical2normal(double theta, double phi, double *n, int lane)
{
    const double arg = (lane & 2) ? phi : theta;
    const int wantCos = lane & 1;
    double y0, y1;
    const int q = (det_rem_pio2(arg, &y0, &y1) + wantCos) & 3;
    const double ks = det_ksin(y0, y1, 1), kc = det_kcos(y0, y1);
    double r = (q & 1) ? kc : ks;
    r = (q & 2) ? -r : r;
    if (arg != arg || arg - arg != 0.0) r = arg - arg; // NaN / inf, as det_sin / det_cos
    const double st = lane_get(r, 0), ct = lane_get(r, 1), sp = lane_get(r, 2), cp = lane_get(r, 3);
    n[0] = st * cp;
    n[1] = st * sp;
    n[2] = ct;
}

template <int NS, bool BYTES, bool ACCR>
__device__ int eval_fitness_parts(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *cbuf,
                                  const WinPix *win, double theta, double phi, double depth, int lane, int part, int nparts,
                                  double *f4, double *w4)
{
    f4[0] = f4[1] = f4[2] = f4[3] = 0;
    w4[0] = w4[1] = w4[2] = w4[3] = 0;
    double n[3];
#if PAIS_WAVE_SINCOS
    wave_spherical2normal(theta, phi, n, lane);
#else
    spherical2normal(theta, phi, n);
#endif
    {
        double on[3] = {ep->optNref[0], ep->optNref[1], ep->optNref[2]};
        if (dot3(n, on) > 0) return 1; // patch.cpp:939
    }
    if (!ep->valid) return 1;          // :952-962, the same for every particle of the run
    if (!(fabs(depth) > 0)) return 1;  // centre == C_ref (or NaN): the reference's projection is 0/0 (:952)
    double center[3];
    for (int i = 0; i < 3; ++i) center[i] = ep->ray[i] * depth + ep->Cref[i]; // :944
    const int M = ep->M, K = ep->K;
    const double s = ep->lodScale;
#if defined(PAIS_EXP_DUP) && PAIS_EXP_DUP == 7
    for (int rep = 0; rep < 2; ++rep) // (measurement build: the normal and the homographies twice, scripts/dup_profile.sh)
#endif
    {
#if defined(PAIS_EXP_DUP) && PAIS_EXP_DUP == 7
        asm volatile("" : "+v"(theta));
        wave_spherical2normal(theta, phi, n, lane);
        for (int i = 0; i < 3; ++i) center[i] = ep->ray[i] * depth + ep->Cref[i];
#endif
        const double d = -dot3(center, n);
        double Mref[9], invH[9], kr[9], kt[3];
        for (int i = 0; i < 9; ++i) kr[i] = ep->KRref[i];
        for (int i = 0; i < 3; ++i) kt[i] = ep->KTref[i];
        plane_matrix(d, s, kr, kt, n, Mref);
        inv3(Mref, invH);
        for (int c = lane; c < M; c += 64) {
            double H[9];
            if (cams[c].cam == ep->refCam) { // :317-320 (a second occurrence of the reference camera)
                H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
            } else {
                double Mc[9];
                for (int i = 0; i < 9; ++i) kr[i] = cams[c].KR[i];
                for (int i = 0; i < 3; ++i) kt[i] = cams[c].KT[i];
                plane_matrix(d, s, kr, kt, n, Mc);
                mul33(Mc, invH, H);
            }
            for (int i = 0; i < 9; ++i) Hbuf[c * PAIS_H_STRIDE + i] = H[i];
            // (pais_ctx_create bounds the blob: levels start below 2^40 elements and are at most 65535 pixels wide)
            Hbuf[c * PAIS_H_STRIDE + 9] = __longlong_as_double((long long)((cams[c].imgOff & 0xFFFFFFFFFFull) | ((uint64_t)(uint32_t)cams[c].w << 40)));
        }
    }
    wave_sync();
#if PAIS_CORNER_FASTPATH
    if (corners_inside(ep, cams, Hbuf, sc.cfg.patchSize, lane))
        return eval_window<NS, false, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win, lane, part, nparts, f4, w4);
#endif
    return eval_window<NS, true, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win, lane, part, nparts, f4, w4);
}

// ---- round 6: the walk of the two-pixel kernels with the taps of the NEXT trip in flight under the tail of this one ----
// A trip of eval_window is: homographies (LDS) -> tap addresses -> 8..12 gathers -> s_waitcnt -> lerps -> mean / deviations / exp
// -> accumulate.  The gathers are L2 hits whose whole latency the wave sat out (SQ_WAIT_ANY 45-50 % of the wave cycles at
// 3 waves / SIMD, VALU issue 63-70 %: profiles/r06_pre_setup_ab.txt).  Rotated: the first camera group of trip t + 1 is
// addressed and requested BEFORE the tail of trip t (its ~100 VALU instructions and LDS round trips then run under the
// gathers), and finished at the top of the next iteration.  The same loads, the same operations on the same operands in the
// same order per value: same bits.  The group is taken with a wave-uniform size (1, 2 or 3 cameras: the first group of the
// partition tap_group's callers use), so one copy of the code serves every camera count.
#ifndef PAIS_PIPE_TAPS
#define PAIS_PIPE_TAPS 0 // measured slower (register spills: profiles/r06_pipe_taps_ab.txt); 1 builds it
#endif
#ifndef PAIS_PIPE_CAP
#define PAIS_PIPE_CAP 3 // cameras of the group in flight under the tail (3: also the triple of a 4-camera patch; 2: pairs and singles only)
#endif
template <int NS, bool BYTES> struct TapPend {
    typename Tap<BYTES>::Row r0[NS][PAIS_PIPE_CAP], r1[NS][PAIS_PIPE_CAP];
    double bx[NS][PAIS_PIPE_CAP], by[NS][PAIS_PIPE_CAP];
};
// G (wave-uniform, 1..3) cameras from c0 on: everything of tap_group up to and including the loads
template <int NS, bool CHECK, bool BYTES>
__device__ __forceinline__ void tap_issue(const DevScene &sc, const EvalCam *cams, const double *Hbuf, int c0, int G, double *x, double *y,
                                          uint32_t *badBits, TapPend<NS, BYTES> &P)
{
#pragma unroll
    for (int q = 0; q < NS; ++q) asm volatile("" : "+v"(x[q]), "+v"(y[q]));
    typedef typename Tap<BYTES>::Row RowTap;
    constexpr uint32_t kElem = Tap<BYTES>::kElem;
    double nx[NS][PAIS_PIPE_CAP], ny[NS][PAIS_PIPE_CAP], w[NS][PAIS_PIPE_CAP], rw[NS][PAIS_PIPE_CAP];
    uint64_t tapWord[PAIS_PIPE_CAP];
#pragma unroll
    for (int u = 0; u < PAIS_PIPE_CAP; ++u) {
        if (u < G) {
            const double2 *H2 = (const double2 *)__builtin_assume_aligned(Hbuf + PAIS_H_STRIDE * (c0 + u), 16);
            const double2 ha = H2[0], hb = H2[1], hc = H2[2], hd = H2[3], he = H2[4];
            const double h0 = ha.x, h1 = ha.y, h2 = hb.x, h3 = hb.y, h4 = hc.x, h5 = hc.y, h6 = hd.x, h7 = hd.y, h8 = he.x;
            tapWord[u] = (uint64_t)__double_as_longlong(he.y);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                w[q][u] = fma(h7, y[q], fma(h6, x[q], h8));
                nx[q][u] = fma(h1, y[q], fma(h0, x[q], h2));
                ny[q][u] = fma(h4, y[q], fma(h3, x[q], h5));
            }
        }
    }
    // one reciprocal per group and pixel: tap_group's statements by group size
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        if (PAIS_PIPE_CAP == 3 && G == 3) {
            constexpr int L = PAIS_PIPE_CAP - 1;
            const double p01 = w[q][0] * w[q][1];
            const double r = rcp_cr(p01 * w[q][L]);
            rw[q][L] = r * p01;
            const double r01 = r * w[q][L];
            rw[q][0] = r01 * w[q][1];
            rw[q][1] = r01 * w[q][0];
        } else if (G == 2) {
            const double r = rcp_cr(w[q][0] * w[q][1]);
            rw[q][0] = r * w[q][1];
            rw[q][1] = r * w[q][0];
        } else {
            rw[q][0] = rcp_cr(w[q][0]);
        }
    }
#pragma unroll
    for (int u = 0; u < PAIS_PIPE_CAP; ++u) {
        if (u < G) {
            const int c = c0 + u;
            int qxmax = 0, qymax = 0;
            if (CHECK) {
                const uint32_t qp = cams[c].qpack;
                qxmax = (int)(qp & 0xffffu);
                qymax = (int)(qp >> 16);
            }
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)tapWord[u]), hi = __builtin_amdgcn_readfirstlane((uint32_t)(tapWord[u] >> 32));
            const unsigned char *base = Tap<BYTES>::blob(sc) + (((uint64_t)(hi & 0xffu) << 32) | lo) * kElem;
            const uint32_t cw = hi >> 8;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const double ix = nx[q][u] * rw[q][u], iy = ny[q][u] * rw[q][u];
                const int qx = (int)ix, qy = (int)iy;
                int px = qx, py = qy;
                if (CHECK) {
                    px = clamp_i32(qx, 2, qxmax);
                    py = clamp_i32(qy, 2, qymax);
                    badBits[q] = badBits[q] | (uint32_t)(px ^ qx) | (uint32_t)(py ^ qy);
                }
                P.bx[q][u] = __builtin_amdgcn_fract(ix);
                P.by[q][u] = __builtin_amdgcn_fract(iy);
                const uint32_t off = (__umul24((uint32_t)py, cw) + (uint32_t)px) * kElem;
                P.r0[q][u] = load_row_at<RowTap>(base, off);
                P.r1[q][u] = load_row_at<RowTap>(base, off + cw * kElem);
            }
        }
    }
}
// ... and the rest of tap_group: lerps, colour rows, running sums (camera order within the pixel as in tap_group)
template <int NS, bool BYTES>
__device__ __forceinline__ void tap_finish(const TapPend<NS, BYTES> &P, double *myc, int c0, int G, double *sum)
{
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < PAIS_PIPE_CAP; ++u) {
            if (u < G) {
                const double col = lerp3(P.r0[q][u], P.r1[q][u], P.bx[q][u], P.by[q][u]);
                myc[((c0 + u) * NS + q) * 64] = col;
                sum[q] += col;
            }
        }
}

template <int NS, bool CHECK, bool BYTES, bool ACCR>
__device__ int eval_window_pipe(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *cbuf,
                                const WinPix *win, int lane, int part, int nparts, double *f4, double *w4)
{
    static_assert(NS == 2, "the rotated walk is the two-pixel kernels' (<= 12 cameras: one sequential colour sum)");
    const int M = __builtin_amdgcn_readfirstlane(ep->M), K = ep->K;
    const int S = sc.cfg.patchSize, S2 = S * S;
    const double a0 = ep->a0, b0 = ep->b0;
    const double invDiffW = uniform_d(1.0 / sc.cfg.diffWeighting);
    const bool useDiff = sc.cfg.adaptiveDifferenceEnable != 0;
    const bool hasRef = ep->hasRef != 0;
    const double invK = 1.0 / (double)K;
    double *myc = cbuf + lane;
    constexpr bool ACCREG = ACCR;
    double accF[4] = {0, 0, 0, 0}, accW[4] = {0, 0, 0, 0};
    double *myacc = cbuf + (size_t)M * NS * 64 + lane;
    if (!ACCREG) {
#pragma unroll
        for (int a = 0; a < 8; ++a) myacc[a * 64] = 0;
    }
    const int adv = 64 * nparts;
    const int qA = adv / S, rA = adv - qA * S;
    int yw = (64 * part + lane) / S, xw = (64 * part + lane) - yw * S;
    // first group of the partition: pairs while 4 or more (or exactly 2) cameras remain, then a triple or a single one
    const int G1 = M == 0 ? 0 : (M == 1 ? 1 : (M == 3 ? (PAIS_PIPE_CAP == 3 ? 3 : 0) : 2)); // (0: nothing in flight, the groups as in eval_window)
    double x[NS], y[NS];
    WinPix wp[NS];
    uint32_t badBits[NS];
    int gi[NS];
    TapPend<NS, BYTES> P;
    int st = part;
    bool have = 64 * st < S2;
#define PAIS_TRIP_HEAD()                                               \
    _Pragma("unroll") for (int q = 0; q < NS; ++q)                    \
    {                                                                  \
        const int stq = st + q * nparts;                               \
        const int k = 64 * stq + lane;                                 \
        wp[q] = win[64 * stq < S2 ? k : (S2 - 1)];                     \
        x[q] = a0 + (double)xw;                                        \
        y[q] = b0 + (double)yw;                                        \
        badBits[q] = 0;                                                \
        gi[q] = stq & 3;                                               \
        xw += rA; yw += qA;                                            \
        yw += (xw >= S) ? 1 : 0;                                       \
        xw -= (xw >= S) ? S : 0;                                       \
    }
    if (have) {
        PAIS_TRIP_HEAD()
        if (G1) tap_issue<NS, CHECK, BYTES>(sc, cams, Hbuf, 0, G1, x, y, badBits, P);
    }
    while (have) {
        double sum[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) sum[q] = hasRef ? wp[q].refCol : 0.0;
        if (G1) tap_finish<NS, BYTES>(P, myc, 0, G1, sum);
        int c0 = G1;
        for (; M - c0 >= 4 || M - c0 == 2; c0 += 2) tap_group<2, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum);
        if (M - c0 == 3) tap_group<3, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum);
        else if (M - c0 == 1) tap_group<1, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum);
        // what the tail of this trip needs, set aside; then the next trip's head and the request of its first group
        double tRef[NS], tStat[NS];
        uint32_t tBad[NS];
        int tGi[NS];
        const int tSt = st;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            tRef[q] = wp[q].refCol;
            tStat[q] = wp[q].wStat;
            tBad[q] = badBits[q];
            tGi[q] = gi[q];
        }
        st += NS * nparts;
        have = 64 * st < S2;
        if (have) {
            PAIS_TRIP_HEAD()
            if (G1) tap_issue<NS, CHECK, BYTES>(sc, cams, Hbuf, 0, G1, x, y, badBits, P);
        }
        // tail of trip tSt (eval_window's statements)
        double mean[NS], sad[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            mean[q] = sum[q] * invK;
            sad[q] = hasRef ? fabs(tRef[q] - mean[q]) : 0.0;
        }
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int q = 0; q < NS; ++q) sad[q] += fabs(myc[(c * NS + q) * 64] - mean[q]);
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (64 * (tSt + q * nparts) >= S2) break; // uniform: the window has no such step
            const bool act = tStat[q] >= 0.0;
            if (CHECK && __any(act && tBad[q] != 0)) return 1; // :1001 -- whole call
            const double sadq = sad[q] * invK;
            double weight = tStat[q];
            if (useDiff) weight *= det_exp_poly(mul_uniform(-(sadq * sadq), invDiffW));
            if (ACCREG) {
#define PAIS_ACC(a)                                           \
    {                                                         \
        accW[a] = act ? (accW[a] + weight) : accW[a];         \
        accF[a] = act ? fma(weight, sadq, accF[a]) : accF[a]; \
    }
                const int ga = __builtin_amdgcn_readfirstlane(tGi[q]);
                if (ga == 0) PAIS_ACC(0) else if (ga == 1) PAIS_ACC(1) else if (ga == 2) PAIS_ACC(2) else PAIS_ACC(3)
#undef PAIS_ACC
            } else {
                double *pa = myacc + tGi[q] * 128;
                const double w0 = pa[64], f0 = pa[0];
                pa[64] = act ? (w0 + weight) : w0;
                pa[0] = act ? fma(weight, sadq, f0) : f0;
            }
        }
    }
#undef PAIS_TRIP_HEAD
    if (nparts == 1) {
        double a8[8], t8[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            a8[a] = ACCREG ? accF[a] : myacc[a * 128];
            a8[4 + a] = ACCREG ? accW[a] : myacc[a * 128 + 64];
        }
        wave_sum8(a8, t8);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f4[a] = t8[a];
            w4[a] = t8[4 + a];
        }
        return 0;
    }
    if (ACCREG) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if ((a - part) % nparts != 0 || a < part) continue;
            f4[a] = wave_sum_x(accF[a]);
            w4[a] = wave_sum_x(accW[a]);
        }
    } else {
        for (int a = part; a < 4; a += nparts) {
            f4[a] = wave_sum_x(myacc[a * 128]);
            w4[a] = wave_sum_x(myacc[a * 128 + 64]);
        }
    }
    return 0;
}

// the window walk of one evaluation (homographies in Hbuf); CHECK: see corners_inside
template <int NS, bool CHECK, bool BYTES, bool ACCR>
__device__ int eval_window(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *cbuf,
                           const WinPix *win, int lane, int part, int nparts, double *f4, double *w4)
{
#if PAIS_PIPE_TAPS
    if constexpr (NS == 2) return eval_window_pipe<NS, CHECK, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win, lane, part, nparts, f4, w4);
#endif
    // (wave-uniform by construction; said to the compiler, which otherwise keeps the camera loops' counters in vector registers and
    //  runs them under exec masks: 3 of the 7 VALU instructions per camera of the deviation loop were the counter)
    const int M = __builtin_amdgcn_readfirstlane(ep->M), K = __builtin_amdgcn_readfirstlane(ep->K);
    const int S = sc.cfg.patchSize, S2 = S * S;
    const double a0 = ep->a0, b0 = ep->b0;
    const double invDiffW = uniform_d(1.0 / sc.cfg.diffWeighting);
    const bool useDiff = sc.cfg.adaptiveDifferenceEnable != 0;
    const bool hasRef = ep->hasRef != 0;
    const double invK = 1.0 / (double)K;
    const bool twoLevel = K >= PAIS_TWO_LEVEL_K;      // (wave-uniform)
    const int hSplit = two_level_split(K, M);
    double *myc = cbuf + lane;
    constexpr bool ACCREG = ACCR;
    double accF[4] = {0, 0, 0, 0}, accW[4] = {0, 0, 0, 0}; // the lane's sub-accumulators (ACCREG)
    double *myacc = cbuf + (size_t)M * NS * 64 + lane;     // [2a] fitness, [2a+1] weight of sub-accumulator a (!ACCREG)
    if (!ACCREG) {
#pragma unroll
        for (int a = 0; a < 8; ++a) myacc[a * 64] = 0;
    }

    // Branch-free over lanes: every lane runs the same straight-line tap code (clamped pixel index / clamped
    // addresses for lanes that have no pixel, a masked pixel or an overflowing tap); only wave-uniform
    // conditions branch.  Contributions are selected at the end.
    // This wave's steps are part, part + nparts, ...: NS of them per iteration.  The lane's window pixel is
    // advanced by 64 * nparts pixels per step without a division.
    const int adv = 64 * nparts;
    const int qA = adv / S, rA = adv - qA * S; // uniform
    int yw = (64 * part + lane) / S, xw = (64 * part + lane) - yw * S;
    for (int st = part; 64 * st < S2; st += NS * nparts) {
        double x[NS], y[NS], sum[NS];
        WinPix wp[NS];
        uint32_t badBits[NS];
        int gi[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int stq = st + q * nparts;
            const int k = 64 * stq + lane;
            // requested before the taps so that the latency hides behind them.  A whole step past the window (uniform:
            // skipped below) re-reads the last one; the padding lanes of the last step are masked entries whose taps are
            // clamped for addressing like any overflowing tap
            wp[q] = win[64 * stq < S2 ? k : (S2 - 1)];
            x[q] = a0 + (double)xw;
            y[q] = b0 + (double)yw;
            badBits[q] = 0; // != 0: some tap of this pixel left [2, w-3) x [2, h-3)
            gi[q] = stq & 3; // canonical sub-accumulator of the step
            xw += rA; yw += qA;
            yw += (xw >= S) ? 1 : 0;
            xw -= (xw >= S) ? S : 0;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) sum[q] = hasRef ? wp[q].refCol : 0.0;
        // (two-level sums, K >= PAIS_TWO_LEVEL_K -- one-pixel kernels only: the two-pixel shapes are chosen for batches of <= 12
        //  cameras.  Groups never straddle hSplit: it is even, the groups are pairs from 0 and one tail group at the end.  When the
        //  walk reaches camera hSplit the first group's sum is set aside and the accumulator restarts from 0: 0 + c == c exactly)
        double sumFirst[NS];
#define PAIS_AT_SPLIT(c)                                                       \
    if (NS == 1 && twoLevel && (c) == hSplit) {                                \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) { sumFirst[q] = sum[q]; sum[q] = 0.0; } \
    }
        int c0 = 0;
#if PAIS_NS1_UNROLL > 1
        // many cameras, one pixel per lane: PAIS_NS1_UNROLL pairs per trip, so that the loads of the later pairs are in flight
        // while the first is interpolated (same groups, same arithmetic: a trip is taken only where the one-by-one rule
        // below would take that many pairs anyway)
        if (NS == 1)
            for (; M - c0 >= 2 * PAIS_NS1_UNROLL + 2 || M - c0 == 2 * PAIS_NS1_UNROLL; c0 += 2 * PAIS_NS1_UNROLL) {
#pragma unroll
                for (int u = 0; u < PAIS_NS1_UNROLL; ++u) {
                    PAIS_AT_SPLIT(c0 + 2 * u)
                    tap_group<2, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0 + 2 * u, x, y, badBits, sum);
                }
            }
#endif
        for (; M - c0 >= 4 || M - c0 == 2; c0 += 2) { // pairs
            PAIS_AT_SPLIT(c0)
            tap_group<2, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum);
        }
        PAIS_AT_SPLIT(c0)
        if (M - c0 == 3) tap_group<3, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum);      // odd count: one triple
        else if (M - c0 == 1) tap_group<1, NS, CHECK, BYTES>(sc, cams, Hbuf, myc, c0, x, y, badBits, sum); // a single camera
#undef PAIS_AT_SPLIT
        // mean and mean absolute deviation of the K colours: one pass over the cameras serves the lane's NS pixels
        double mean[NS], sad[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (NS == 1 && twoLevel) sum[q] = sumFirst[q] + sum[q];
            mean[q] = sum[q] * invK;
            sad[q] = hasRef ? fabs(wp[q].refCol - mean[q]) : 0.0;
        }
        for (int c = 0; c < hSplit; ++c) {
#pragma unroll
            for (int q = 0; q < NS; ++q) sad[q] += fabs(myc[(c * NS + q) * 64] - mean[q]);
        }
        if (NS == 1 && twoLevel) {
            double sadB[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) sadB[q] = 0.0;
            for (int c = hSplit; c < M; ++c) {
#pragma unroll
                for (int q = 0; q < NS; ++q) sadB[q] += fabs(myc[(c * NS + q) * 64] - mean[q]);
            }
#pragma unroll
            for (int q = 0; q < NS; ++q) sad[q] += sadB[q];
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (64 * (st + q * nparts) >= S2) break; // uniform: the window has no such step
            const bool act = wp[q].wStat >= 0.0;
            if (CHECK && __any(act && badBits[q] != 0)) return 1; // :1001 -- whole call
            const double sadq = sad[q] * invK;
            double weight = wp[q].wStat;
            if (useDiff) weight *= det_exp_poly(mul_uniform(-(sadq * sadq), invDiffW));
            if (ACCREG) {
                // the sub-accumulator index is wave-uniform: a scalar branch selects the registers
#define PAIS_ACC(a)                                           \
    {                                                         \
        accW[a] = act ? (accW[a] + weight) : accW[a];         \
        accF[a] = act ? fma(weight, sadq, accF[a]) : accF[a]; \
    }
                const int ga = __builtin_amdgcn_readfirstlane(gi[q]);
                if (ga == 0) PAIS_ACC(0) else if (ga == 1) PAIS_ACC(1) else if (ga == 2) PAIS_ACC(2) else PAIS_ACC(3)
#undef PAIS_ACC
            } else {
                double *pa = myacc + gi[q] * 128;
                const double w0 = pa[64], f0 = pa[0];
                pa[64] = act ? (w0 + weight) : w0;
                pa[0] = act ? fma(weight, sadq, f0) : f0;
            }
        }
    }
    // one wave computes all four sub-accumulators: their eight butterflies as one (wave_sum8: same pairs, same bits)
#if PAIS_WAVE_SUM8
    if (nparts == 1) {
        double a8[8], t8[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            a8[a] = ACCREG ? accF[a] : myacc[a * 128];
            a8[4 + a] = ACCREG ? accW[a] : myacc[a * 128 + 64];
        }
        wave_sum8(a8, t8);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f4[a] = t8[a];
            w4[a] = t8[4 + a];
        }
        return 0;
    }
    if (nparts == 2) { // sub-accumulators part and part + 2
        double a4[4], t4[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // (part is wave-uniform: the selects below are scalar)
            const double fA = ACCREG ? (part == 0 ? accF[2 * k] : accF[2 * k + 1]) : myacc[(part + 2 * k) * 128];
            const double wA = ACCREG ? (part == 0 ? accW[2 * k] : accW[2 * k + 1]) : myacc[(part + 2 * k) * 128 + 64];
            a4[k] = fA;
            a4[2 + k] = wA;
        }
        wave_sum4(a4, t4);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if ((a & 1) == part) {
                f4[a] = t4[a >> 1];
                w4[a] = t4[2 + (a >> 1)];
            }
        return 0;
    }
    if (nparts == 4) { // sub-accumulator part
        double a2[2], t2[2];
        a2[0] = ACCREG ? (part == 0 ? accF[0] : part == 1 ? accF[1] : part == 2 ? accF[2] : accF[3]) : myacc[part * 128];
        a2[1] = ACCREG ? (part == 0 ? accW[0] : part == 1 ? accW[1] : part == 2 ? accW[2] : accW[3]) : myacc[part * 128 + 64];
        wave_sum2(a2, t2);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a == part) {
                f4[a] = t2[0];
                w4[a] = t2[1];
            }
        return 0;
    }
#endif
    // butterflies only for this wave's sub-accumulators (uniform conditions)
    if (ACCREG) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if ((a - part) % nparts != 0 || a < part) continue;
            f4[a] = wave_sum_x(accF[a]);
            w4[a] = wave_sum_x(accW[a]);
        }
    } else {
        for (int a = part; a < 4; a += nparts) {
            f4[a] = wave_sum_x(myacc[a * 128]);
            w4[a] = wave_sum_x(myacc[a * 128 + 64]);
        }
    }
    return 0;
}
__device__ __forceinline__ double combine_parts(const double *f4, const double *w4)
{
    const double F = ((f4[0] + f4[1]) + f4[2]) + f4[3];
    const double W = ((w4[0] + w4[1]) + w4[2]) + w4[3];
    return F / W; // NaN when every pixel was masked, as in the reference
}

// copies the candidate's [EvalPatch][EvalCam x M] block (global, prepared by build_eval_block) into this wave's LDS
__device__ __forceinline__ void stage_eval_block(unsigned char *smem, const uint64_t *src, int nw, int lane, uint64_t v0, uint64_t v1)
{
    uint64_t *dst = (uint64_t *)smem;
    if (lane < nw) dst[lane] = v0;
    if (lane + 64 < nw) dst[lane + 64] = v1;
    for (int q = lane + 128; q < nw; q += 64) dst[q] = src[q];
}
