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
//   2. every wave lays the per-camera tiles out in the tile area (camera c in lane c: an exclusive scan of the sizes; the
//      same layout in every wave, so no second barrier) and notes each camera's tile in its homography record;
//   3. the waves issue the copy of the tiles global -> LDS as LDS-DMA (global_load_lds_dword: no registers; every
//      instruction of the strip in flight at once) -- the only global image traffic; one wait + barrier; then
//   4. every wave walks the strip's steps for its particle: the taps are byte reads at (py - y0) * tw + (px - x0).
// The colours of a pixel stay in REGISTERS (camera pairs statically unrolled, the odd tail's group behind them), so the
// workgroup's LDS is the tiles + 2.5 KB of homographies per wave: 8 waves per CU at <= 256 VGPRs.  A lane carries TWO
// window pixels (consecutive steps) through every camera group: the wave-uniform operands of a camera -- its homography
// and its tile -- are LDS reads (ds_read_b128, 4 LDS cycles each whatever the lanes read), the LDS pipe is shared by the
// CU's four SIMDs, and those reads keep it as busy as the VALUs: the LDS pipe, not the VALU, bounds the walk (measured:
// dropping ONE 8-byte wave-uniform read per camera and pixel -- the tile word, now in the padding of the homography record
// -- shortened the walk by 15 %).  A camera costs five 16-byte reads per two pixels (one pixel beyond 32 cameras).
// Same arithmetic, same operation order and same reduction shape as eval_window<1, false, true, true>: identical bits
// (tests/test_gpu_parity.py: test_dome_radius25_many_cameras).
//
// What does NOT go through the tiles, exactly as before:
//   * a particle whose window corners do not map inside every image with one sign of the denominator (corners_inside):
//     it is flagged "pending" and the k_pso_eval2 launch that follows evaluates it with the checked walk;
//   * a camera whose tile does not fit the tile area any more (huge magnification): tapped from global memory.
#pragma once

#ifndef TILE_WAVES
#define TILE_WAVES 8              // waves (= particles) of a workgroup
#endif
#ifndef TILE_EXP_DUP
#define TILE_EXP_DUP 0            // measurement builds (scripts/dome_dup_profile.sh): ONE component of the walk executed twice, same results:
#endif                            // 1 homography reads, 2 byte taps, 5 WinPix loads (3 a whole camera group, 4 the per-pixel tail: round 5's builds, retired
                                  // with the two-group registers of round 6; profiles/r05_dome_tile_dup_profile.txt has their figures)
#ifndef TILE_EXP_SKIP
#define TILE_EXP_SKIP 0           // measurement builds, WRONG results (timing of the first seed pass only, scripts/dome_skip_profile.sh):
#endif                            // 1 four of the five homography reads per camera replaced by constants, 2 the byte taps, 3 the exp of the tail
#ifndef TILE_STAGGER
#define TILE_STAGGER 0           // see the walk
#endif
#ifndef TILE_WGS_PER_CU
#define TILE_WGS_PER_CU 1         // workgroups that share a CU's 160 KB of LDS (each gets 160 / TILE_WGS_PER_CU KB)
#endif
#define TILE_MAX_CAMS PAIS_MAX_VIS // cameras of a candidate.  Two instantiations: NS = 2 pixels per lane, 16 camera pairs in registers
                                  // (M <= 32 tapped cameras); NS = 1, 32 pairs (M <= 64)
#ifndef TILE_STRIP_STEPS
#define TILE_STRIP_STEPS 12       // 64-pixel steps per strip, even (r = 25: 41 steps -> strips of 12, 12, 12, 5: ~15 window rows)
#endif
#define TILE_DOUBLE_BUFFER 0      // (a variant with two half-size tile areas -- the next strip staged while this one is walked -- was
                                  //  measured slower, see the header; the half / halfBytes arithmetic of the layout is what is left of it)

// tile of a camera in a strip, as the taps read it: two ints in the 10th (padding) double of the camera's homography record
//   base : byte index in the tile area of image pixel (0, 0): off - y0 * tw - x0
//   tw   : row stride of the tile in bytes (multiple of 4); 0: not staged, the camera is tapped in global memory
struct TileBox { int32_t xmin, ymin, xmax, ymax; };

__host__ __device__ inline size_t tile_fixed_lds_bytes(int Kmax)
{
    size_t b = eval_block_bytes(Kmax);                                   // EvalPatch + EvalCam[Kmax], shared by the waves
    b += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE_WAVES;     // homographies, per wave
    b += 2 * sizeof(TileBox) * (size_t)Kmax + 64;                        // boxes (two sets), flags
    return (b + 15) & ~(size_t)15;
}

// one camera group of the lane's two window pixels from the tiles: the statements of tap_group<G, 1, false, true> per
// pixel, with LDS rows
// PAIS_TILE_SCALAR_H (round 4): where a wave's homographies come from in the walk.
//   0  LDS (ds_read_b128 x 5 per camera and trip: wave-uniform operands, 1 KB through the LDS data path each);
//   1  the one-pixel instantiation reads them through the SCALAR cache: every wave also writes its homographies to a slot of
//      a global scratch, and the walk loads them with s_load (constant address space: SGPR operands of the fma's, no LDS
//      cycles); only the camera's tile word (8 bytes, rewritten per strip) stays an LDS read.  The LDS pipe, shared by the
//      CU's four SIMDs, is what the one-pixel walk saturates (one homography read per camera and PIXEL there);
//   2  both instantiations.
// MEASURED (round 4, profiles/r04_tile_scalar_h_ab.txt; dome seeds + 10 rounds, one box, alternating; same cloud hash, the
// verify mode of test_dome_radius25_many_cameras green): 1 is SLOWER -- 2 797 / 2 801 ms against 2 409 / 2 407 ms per
// reconstruction (-14 %).  The LDS reads do disappear (ISA of the one-pixel kernel: 16 ds_read_b128 left of 180, 66
// s_load_dwordx16 + 72 s_load_dwordx2 instead), but a wave re-reads its 34-43 homographies (2.7-3.4 KB) once per 64-pixel
// step, eight waves of a workgroup hold 22-27 KB of them, and the scalar data cache is 16 KB: the loads go to L2 on every step,
// and the constant-bus limit of gfx9 (one SGPR operand per VALU instruction) costs ~9 v_mov per camera.  Default 0; the
// variant stays selectable (-DPAIS_TILE_SCALAR_H=1).  (PAIS_TILE_SCALAR_H itself is defined in pais_internal.h: the host
// only allocates the scratch when it is on.)
typedef const double __attribute__((address_space(4))) *TileHS;
template <int G, int NS, bool SH>
__device__ __forceinline__ void tile_tap_group(const DevScene &sc, const EvalCam *cams, const unsigned char *tiles,
                                               const double *Hbuf, TileHS hs, int c0, double *x, double *y, double (*col)[NS], double *sum)
{
#pragma unroll
    for (int q = 0; q < NS; ++q) asm volatile("" : "+v"(x[q]), "+v"(y[q]));
    double nx[NS][G], ny[NS][G], w[NS][G], rw[NS][G];
    int tbase[G], ttw[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        double2 ha, hb, hc, hd, he;
        if (SH) {
            TileHS h = hs + PAIS_H_STRIDE * (c0 + u); // wave-uniform address in the constant address space: s_load
            ha.x = h[0]; ha.y = h[1]; hb.x = h[2]; hb.y = h[3]; hc.x = h[4]; hc.y = h[5]; hd.x = h[6]; hd.y = h[7]; he.x = h[8];
            he.y = Hbuf[PAIS_H_STRIDE * (c0 + u) + 9]; // the tile word of this strip: one 8-byte LDS read
        } else {
            const double2 *H2 = (const double2 *)__builtin_assume_aligned(Hbuf + PAIS_H_STRIDE * (c0 + u), 16);
#if TILE_EXP_SKIP == 1
            he = H2[4];
            ha.x = 1.0; ha.y = 0.0; hb.x = 0.25 * he.x; hb.y = 0.0; hc.x = 1.0; hc.y = 0.125 * he.x; hd.x = 0.0; hd.y = 0.0;
#else
            ha = H2[0]; hb = H2[1]; hc = H2[2]; hd = H2[3]; he = H2[4];
#endif
#if TILE_EXP_DUP == 1
            {
                const double2 *H3 = H2;
                asm volatile("" : "+v"(H3));
                const double2 t0 = H3[0], t1 = H3[1], t2 = H3[2], t3_ = H3[3], t4 = H3[4];
                asm volatile("" ::"v"(t0.x), "v"(t0.y), "v"(t1.x), "v"(t1.y), "v"(t2.x), "v"(t2.y), "v"(t3_.x), "v"(t3_.y), "v"(t4.x), "v"(t4.y));
            }
#endif
        }
        // (the camera's tile rides in the padding of its homography record: no read of its own)
        tbase[u] = __double2loint(he.y);
        ttw[u] = __builtin_amdgcn_readfirstlane(__double2hiint(he.y));
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            w[q][u] = fma(hd.y, y[q], fma(hd.x, x[q], he.x));
            nx[q][u] = fma(ha.y, y[q], fma(ha.x, x[q], hb.x));
            ny[q][u] = fma(hc.x, y[q], fma(hb.y, x[q], hc.y));
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        if (G == 3) {
            const double p01 = w[q][0] * w[q][G > 1 ? 1 : 0];
            const double r = rcp_cr(p01 * w[q][G - 1]);
            rw[q][G - 1] = r * p01;
            const double r01 = r * w[q][G - 1];
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
    // a row of a tap = two adjacent bytes; from the tiles they are read as BYTES: an unaligned 2-byte LDS read stalls the LDS
    // pipe (SQ_LDS_UNALIGNED_STALL: 87 % of the LDS cycles of the first version of this kernel)
    int a0[NS][G], b0[NS][G], a1[NS][G], b1[NS][G];
    double bx[NS][G], by[NS][G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        int px[NS], py[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const double ix = nx[q][u] * rw[q][u], iy = ny[q][u] * rw[q][u];
            px[q] = (int)ix;
            py[q] = (int)iy;
            bx[q][u] = __builtin_amdgcn_fract(ix);
            by[q][u] = __builtin_amdgcn_fract(iy);
        }
#if TILE_EXP_SKIP == 1
        if (ttw[u] == 0) { ttw[u] = 64; tbase[u] = 0; } // (timing build: no global-memory taps at coordinates that mean nothing)
#endif
        if (ttw[u] != 0) { // wave-uniform: the camera's tile is staged
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                // (four BYTE reads: the right-hand neighbours go through an opaque copy of the address, or the compiler fuses
                // each pair into one 2-byte read again)
                const uint32_t a = (uint32_t)(tbase[u] + py[q] * ttw[u] + px[q]);
                uint32_t ar = a + 1;
                asm volatile("" : "+v"(ar));
#if TILE_EXP_SKIP == 2
                a0[q][u] = a & 255; b0[q][u] = ar & 255; a1[q][u] = (a >> 3) & 255; b1[q][u] = (ar >> 5) & 255;
#else
                a0[q][u] = tiles[a]; b0[q][u] = tiles[ar];
                a1[q][u] = tiles[a + (uint32_t)ttw[u]]; b1[q][u] = tiles[ar + (uint32_t)ttw[u]];
#endif
#if TILE_EXP_DUP == 2
                {
                    uint32_t a2 = a, ar2 = ar;
                    asm volatile("" : "+v"(a2), "+v"(ar2));
                    const int e0 = tiles[a2], e1 = tiles[ar2], e2 = tiles[a2 + (uint32_t)ttw[u]], e3 = tiles[ar2 + (uint32_t)ttw[u]];
                    asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(e3));
                }
#endif
            }
        } else {
            TapInfo ti;
            __builtin_memcpy(&ti, __builtin_assume_aligned(&cams[c0 + u].imgOff, 16), sizeof(ti));
            const unsigned char *lvl = sc.imgBlob + ti.imgOff;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const uint32_t off = (uint32_t)py[q] * (uint32_t)ti.w + (uint32_t)px[q];
                const uint16_t r0 = load_row_at<uint16_t>(lvl, off), r1 = load_row_at<uint16_t>(lvl, off + (uint32_t)ti.w);
                a0[q][u] = r0 & 0xff; b0[q][u] = r0 >> 8;
                a1[q][u] = r1 & 0xff; b1[q][u] = r1 >> 8;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < G; ++u) {
            col[u][q] = lerp3((double)a0[q][u], (double)(b0[q][u] - a0[q][u]), (double)a1[q][u], (double)(b1[q][u] - a1[q][u]), bx[q][u], by[q][u]);
            sum[q] += col[u][q];
        }
}

// The evaluation launch of many-camera batches.  Grid: candidates x particle groups of TILE_WAVES; workgroup of TILE_WAVES
// waves.  Writes A.fit[i], or flags the particle pending (A.part[i][0] = 1) for the pending-only k_pso_eval2 launch behind it.
template <int NS, int NP>
__global__ __launch_bounds__(64 * TILE_WAVES, 2) void k_pso_tile(DevScene sc, unsigned char *states, int n, int Nmax, int Kmax,
                                                                const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win,
                                                                int tileBytes, int groups, int stripSteps, unsigned long long *dbg,
                                                                double *hscr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool SH = (PAIS_TILE_SCALAR_H == 2) || (PAIS_TILE_SCALAR_H == 1 && NS == 1);
    // this wave's slot of the homography scratch (SH): Kmax records of PAIS_H_STRIDE doubles
    double *hslot = hscr + ((size_t)blockIdx.x * TILE_WAVES + (size_t)wave) * (size_t)Kmax * PAIS_H_STRIDE;
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    size_t o = eval_block_bytes(Kmax);
    double *Hbuf = (double *)(smem + o) + (size_t)wave * Kmax * PAIS_H_STRIDE; o += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE_WAVES;
    TileBox *boxAll = (TileBox *)(smem + o);                                    o += 2 * sizeof(TileBox) * (size_t)Kmax; // [2][Kmax]
    int *flags = (int *)(smem + o);                                             // [0]: some particle of the group walks the tiles
    unsigned char *tiles = smem + tile_fixed_lds_bytes(Kmax);
    const int halfBytes = TILE_DOUBLE_BUFFER ? ((tileBytes / 2) & ~15) : tileBytes; // (two halves: the strip being walked, the strip being staged)
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
            for (int q = threadIdx.x; q < 2 * Kmax; q += 64 * TILE_WAVES) boxAll[q] = TileBox{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
            if (threadIdx.x == 0) flags[0] = 0;
        }
        __syncthreads();
        const int M = ep->M, K = ep->K;
        const WinPix *wbase = win + (size_t)c * WS;
        const int nPairs = (M >= 2) ? ((M & 1) ? (M - 3) / 2 : M / 2) : 0; // cameras 0 .. 2 nPairs - 1 in pairs, then a tail of 3, 1 or 0
        const int tail0 = 2 * nPairs, nTail = M - tail0;
        const bool twoLevel = K >= PAIS_TWO_LEVEL_K;
        const int pA = twoLevel ? two_level_split(K, M) / 2 : nPairs, pB = nPairs - pA; // pairs of the first / second group

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
                    if (SH)
                        for (int q = 0; q < 9; ++q) hslot[cc * PAIS_H_STRIDE + q] = H[q];
                }
                wave_sync();
                if (!corners_inside(ep, cams, Hbuf, S, lane)) state = 2;
                // (a group of more pairs than this instantiation's registers hold -- a patch of exactly 4 HP cameras none of which is
                //  the reference camera: left to the checked walk like a particle that grazes an image border)
                if (pA > NP / 2 || pB > NP / 2) state = 2;
            }
            if (lane == 0) {
                if (state == 1) A.fit[i] = DBL_MAX;
                A.part[i][0] = (state == 2) ? 1.0 : 0.0;
                if (state == 0) flags[0] = 1;
                if (dbg) atomicAdd(&dbg[state], 1ULL); // [0] particles through the tiles, [1] DBL_MAX, [2] pending
            }
        }
        __syncthreads();
        if (!flags[0]) continue; // nobody walks the tiles (uniform)

        // SH: the homographies this wave has just stored are read back through the scalar cache.  They are at the device's L2
        // once the stores have been acknowledged (vmcnt); lines of this slot that the scalar cache still holds from the
        // previous task are dropped (s_dcache_inv); the pointer is re-defined opaquely so that no load through it can be moved
        // above this point (loads from the constant address space are otherwise free to move over stores)
        TileHS hs = nullptr;
        if (SH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)hslot & 0xffffffffu));
            unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)hslot >> 32));
            asm volatile("" : "+s"(lo), "+s"(hi)::"memory");
            hs = (TileHS)(((uintptr_t)hi << 32) | (uintptr_t)lo);
        }
        // (wave-uniform scalars of the walk live in SGPRs: as VGPR values they were four of the registers the walk spilled)
        const double a0 = uniform_d(ep->a0), b0 = uniform_d(ep->b0);
        const double invDiffW = uniform_d(1.0 / sc.cfg.diffWeighting);
        const bool useDiff = sc.cfg.adaptiveDifferenceEnable != 0;
        const bool hasRef = ep->hasRef != 0;
        const double invK = uniform_d(1.0 / (double)K);
        double accF[4] = {0, 0, 0, 0}, accW[4] = {0, 0, 0, 0};

        // stage(strip, half): boxes -> layout -> LDS-DMA of the strip's tiles into `half`.  Two workgroup barriers inside;
        // every wave calls it with the same arguments.
        auto stage = [&](int s0, int half, int par) {
            const int s1 = min(s0 + stripSteps, nSteps);
            TileBox *box = boxAll + par * Kmax; // (two sets, used alternately: this strip's was cleared during the previous strip's layout)
            // ---- 1. bounding boxes of the strip's rectangle (full window rows ya .. yb) in every camera
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
            __syncthreads(); // the boxes are complete -- and every wave has finished walking the strip before this one
            // ---- 2. layout, by EVERY wave alike (camera cc in lane cc, M <= 64; no second barrier for it): one pixel of margin
            // against the rounding of the corner quotients, +1 for the bilinear neighbour
            int x0 = 0, y0 = 0, tw = 0, th = 0;
            if (lane < M && box[lane].xmax >= box[lane].xmin) {
                const int lw = cams[lane].w, lh = cams[lane].h;
                x0 = max(box[lane].xmin - 1, 0); y0 = max(box[lane].ymin - 1, 0);
                const int x1 = min(box[lane].xmax + 2, lw - 1), y1 = min(box[lane].ymax + 2, lh - 1);
                tw = max(((x1 - x0 + 1) + 3) & ~3, 8); // (>= 2 dwords: the copy's division by tw / 4 is a multiplication by
                                                        //  2^32 / (tw / 4), which a one-dword row would overflow -- a single
                                                        //  window row CAN map onto one image column)
                th = y1 - y0 + 1;
            }
            // (a footprint is bounded only by its level -- up to 65535^2 bytes -- and the scan below adds 64 of them: sizes are
            //  saturated at halfBytes + 1, so that neither the product nor the sum can wrap; every camera from the first one
            //  that does not fit on is uniformly "not staged")
            const int szFull = (th > 0 && tw > (halfBytes + 1) / th) ? halfBytes + 1 : tw * th;
            const int sz = min(szFull, halfBytes + 1);
            int incl = sz;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const int up = __shfl_up(incl, m, 64);
                incl = min(incl + ((lane >= m) ? up : 0), halfBytes + 1);
            }
            const int off = half * halfBytes + incl - sz;
            const bool fits = sz > 0 && incl <= halfBytes;
            if (!fits) { tw = 0; th = 0; }
            if (lane < M) {
                // the tile of camera `lane` as the taps read it -- base = byte index in the tile area of image pixel (0, 0), row
                // stride (0: not staged) -- into the padding double of this wave's homography record of the camera
                Hbuf[lane * PAIS_H_STRIDE + 9] = __hiloint2double(tw, fits ? (off - y0 * tw - x0) : 0);
                if (wave == 0) {
                    boxAll[(par ^ 1) * Kmax + lane] = TileBox{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
                    if (dbg && sz > 0) atomicAdd(&dbg[fits ? 3 : 4], 1ULL); // [3] tiles staged, [4] cameras left in global memory
                    if (dbg && fits) atomicAdd(&dbg[5], (unsigned long long)sz); // [5] bytes staged
                }
            }
            // ---- 3. copy as LDS-DMA: cameras dealt to the waves; a tile is th * tw / 4 consecutive dwords of LDS, dword j of it
            // is image byte (y0 + j / twd) * w + x0 + 4 (j % twd); one instruction moves 64 of them
            int inFlight = 0; // LDS-DMA instructions of this wave not waited for yet (the counter behind s_waitcnt vmcnt has 6 bits)
            for (int cc = wave; cc < M; cc += TILE_WAVES) {
                const int ctw = __shfl(tw, cc, 64), cth = __shfl(th, cc, 64);
                const int cx0 = __shfl(x0, cc, 64), cy0 = __shfl(y0, cc, 64), coff = __shfl(off, cc, 64);
                if (ctw == 0) continue;
                const uint32_t twd = (uint32_t)ctw >> 2, J = twd * (uint32_t)cth;
                const int need = (int)((J + 63) >> 6);
                if (inFlight + need > 48) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    inFlight = 0;
                }
                inFlight += need;
                const uint32_t magic = 0xFFFFFFFFu / twd + 1; // j / twd == umulhi(j, magic) for j * twd < 2^32
                const unsigned char *lvl = sc.imgBlob + cams[cc].imgOff + (size_t)(uint32_t)cy0 * (uint32_t)cams[cc].w + (uint32_t)cx0;
                const uint32_t lw = (uint32_t)cams[cc].w;
                const unsigned ldsTile = (unsigned)(uintptr_t)(tiles + coff);
                for (uint32_t j0 = 0; j0 < J; j0 += 64) {
                    const uint32_t j = j0 + lane;
                    if (j < J) {
                        const uint32_t row = __umulhi(j, magic), d = j - row * twd;
                        const unsigned char *gsrc = lvl + (size_t)row * lw + 4 * d;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsTile + 4 * j0);
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
                    }
                }
            }
        };

        const unsigned long long tc0 = dbg ? __builtin_readcyclecounter() : 0;
        unsigned long long tWalk = 0;
        int sIdx = 0;
        for (int s0 = 0; s0 < nSteps; s0 += stripSteps, ++sIdx) {
            const int s1 = min(s0 + stripSteps, nSteps);
            const int half = 0;
            stage(s0, 0, sIdx & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's LDS-DMA has landed ...
            __syncthreads();                                  // ... everybody's has
            const unsigned long long tc3 = dbg ? __builtin_readcyclecounter() : 0;
#if TILE_STAGGER
            // (experiment, profiles/r05_dome_tile_stagger_ab.txt: the eight waves leave the barrier in lockstep -- the same LDS reads at
            //  the same time, then the same arithmetic at the same time; waves 1 .. 7 start the walk 64 * TILE_STAGGER * wave cycles late)
            switch (wave) {
            case 1: __builtin_amdgcn_s_sleep(1 * TILE_STAGGER); break;
            case 2: __builtin_amdgcn_s_sleep(2 * TILE_STAGGER); break;
            case 3: __builtin_amdgcn_s_sleep(3 * TILE_STAGGER); break;
            case 4: __builtin_amdgcn_s_sleep(4 * TILE_STAGGER); break;
            case 5: __builtin_amdgcn_s_sleep(5 * TILE_STAGGER); break;
            case 6: __builtin_amdgcn_s_sleep(6 * TILE_STAGGER); break;
            case 7: __builtin_amdgcn_s_sleep(7 * TILE_STAGGER); break;
            default: break;
            }
#endif
            // ---- 4. the strip's steps for this wave's particle, NS steps (NS pixels per lane) per trip
            if (state == 0) {
                const int kpix = 64 * s0 + lane;
                int yw = kpix / S, xw = kpix - yw * S;
                const int qA = 64 / S, rA = 64 - qA * S;
                for (int st = s0; st < s1; st += NS) {
                    double x[NS], y[NS], sum[NS];
                    WinPix wp[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        const int stq = st + q;
                        // (a step past the window -- uniform, skipped below -- re-reads the last entry; the padding lanes of
                        // the last step are masked entries)
                        wp[q] = wbase[stq < nSteps ? (64 * stq + lane) : (S2 - 1)];
#if TILE_EXP_DUP == 5
                        {
                            const WinPix *wb2 = wbase;
                            asm volatile("" : "+v"(wb2));
                            const WinPix w2 = wb2[stq < nSteps ? (64 * stq + lane) : (S2 - 1)];
                            asm volatile("" ::"v"(w2.refCol), "v"(w2.wStat));
                        }
#endif
                        x[q] = a0 + (double)xw;
                        y[q] = b0 + (double)yw;
                        xw += rA; yw += qA;
                        yw += (xw >= S) ? 1 : 0;
                        xw -= (xw >= S) ? S : 0;
                        sum[q] = hasRef ? wp[q].refCol : 0.0;
                    }
                    // lanes without a pixel (padding of the last step, the step past the window) tap the strip's first pixel:
                    // an address inside the tiles
                    {
                        const double xs = a0 + (double)((64 * s0) % S), ys = b0 + (double)((64 * s0) / S);
#pragma unroll
                        for (int q = 0; q < NS; ++q) {
                            const bool nopix = 64 * (st + q) + lane >= S2 || st + q >= s1;
                            x[q] = nopix ? xs : x[q];
                            y[q] = nopix ? ys : y[q];
                        }
                    }
                    // The colours of the lane's pixels in registers, in the TWO groups of the kernel arithmetic (pais_eval.hpp, PAIS_TWO_LEVEL_K:
                    // from 13 cameras on the reference colour + the first hSplit cameras are summed as one group, the rest as another; with
                    // fewer cameras everything is the first group): group A = pairs 0 .. pA - 1, group B = pairs pA .. nPairs - 1, each with
                    // its own statically indexed registers and its own running sum -- no test between two tap groups, no select.
                    constexpr int HP = NP / 2; // pairs per group: pA = (M + 4) / 4 <= HP and nPairs - pA <= HP for every M < 4 HP (else: pending, above)
                    double colA[2 * HP][NS], colB[2 * HP][NS], t3[3][NS];
                    double sumB[NS], sumTail[NS]; // (sum[] is group A's, started from the reference colour above; the tail's is formed below)
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        t3[0][q] = t3[1][q] = t3[2][q] = 0;
                        sumB[q] = 0;
                        sumTail[q] = 0;
                    }
#pragma unroll
                    for (int u = 0; u < HP; ++u) {
                        if (u < pA) {
                            tile_tap_group<2, NS, SH>(sc, cams, tiles, Hbuf, hs, 2 * u, x, y, &colA[2 * u], sum);
                        } else {
#pragma unroll
                            for (int q = 0; q < NS; ++q) colA[2 * u][q] = colA[2 * u + 1][q] = 0;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < HP; ++u) {
                        if (u < pB) {
                            tile_tap_group<2, NS, SH>(sc, cams, tiles, Hbuf, hs, 2 * (pA + u), x, y, &colB[2 * u], sumB);
                        } else {
#pragma unroll
                            for (int q = 0; q < NS; ++q) colB[2 * u][q] = colB[2 * u + 1][q] = 0;
                        }
                    }
                    if (nTail == 3) tile_tap_group<3, NS, SH>(sc, cams, tiles, Hbuf, hs, tail0, x, y, t3, sumTail);
                    else if (nTail == 1) tile_tap_group<1, NS, SH>(sc, cams, tiles, Hbuf, hs, tail0, x, y, t3, sumTail);
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        if (st + q >= s1) break; // uniform: the strip (the window) has no such step
                        // the tail group continues the LAST group's sum (camIdx order): B's with two groups, else A's
                        double sA = sum[q], sB = sumB[q];
                        if (twoLevel) {
                            if (nTail >= 1) sB += t3[0][q];
                            if (nTail == 3) {
                                sB += t3[1][q];
                                sB += t3[2][q];
                            }
                            sA += sB; // (first group + second group)
                        } else {
                            if (nTail >= 1) sA += t3[0][q];
                            if (nTail == 3) {
                                sA += t3[1][q];
                                sA += t3[2][q];
                            }
                        }
                        const double mean = sA * invK;
                        double sad = hasRef ? fabs(wp[q].refCol - mean) : 0.0, sadB = 0.0;
#pragma unroll
                        for (int u = 0; u < HP; ++u) {
                            if (u < pA) {
                                sad += fabs(colA[2 * u][q] - mean);
                                sad += fabs(colA[2 * u + 1][q] - mean);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < HP; ++u) {
                            if (u < pB) {
                                sadB += fabs(colB[2 * u][q] - mean);
                                sadB += fabs(colB[2 * u + 1][q] - mean);
                            }
                        }
                        if (twoLevel) {
                            if (nTail >= 1) sadB += fabs(t3[0][q] - mean);
                            if (nTail == 3) {
                                sadB += fabs(t3[1][q] - mean);
                                sadB += fabs(t3[2][q] - mean);
                            }
                            sad += sadB;
                        } else {
                            if (nTail >= 1) sad += fabs(t3[0][q] - mean);
                            if (nTail == 3) {
                                sad += fabs(t3[1][q] - mean);
                                sad += fabs(t3[2][q] - mean);
                            }
                        }
                        const bool act = wp[q].wStat >= 0.0;
                        const double sadq = sad * invK;
                        double weight = wp[q].wStat;
#if TILE_EXP_SKIP == 3
                        if (useDiff) weight *= mul_uniform(-(sadq * sadq), invDiffW);
#else
                        if (useDiff) weight *= det_exp_poly(mul_uniform(-(sadq * sadq), invDiffW));
#endif
                        const int ga = (st + q) & 3; // canonical sub-accumulator of the step (uniform)
#define PAIS_TACC(a)                                          \
    {                                                         \
        accW[a] = act ? (accW[a] + weight) : accW[a];         \
        accF[a] = act ? fma(weight, sadq, accF[a]) : accF[a]; \
    }
                        if (ga == 0) PAIS_TACC(0) else if (ga == 1) PAIS_TACC(1) else if (ga == 2) PAIS_TACC(2) else PAIS_TACC(3)
#undef PAIS_TACC
                    }
                }
            }
            if (dbg) tWalk += __builtin_readcyclecounter() - tc3;
        }
        if (dbg && threadIdx.x == 0) {
            atomicAdd(&dbg[6], __builtin_readcyclecounter() - tc0 - tWalk); // staging (boxes, layout, DMA issue, barriers) ...
            atomicAdd(&dbg[9], tWalk);                                      // ... and the walks of wave 0
        }
        if (state == 0) {
            double f4[4], w4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f4[a] = wave_sum_x(accF[a]);
                w4[a] = wave_sum_x(accW[a]);
            }
            if (lane == 0) A.fit[i] = combine_parts(f4, w4);
        }
    }
}
