// pais_tile2.hpp -- PAIS::getFitness (TMVS/mvs/patch.cpp:914-1047) for patches seen by MANY cameras: the LDS-tile kernel of
// pais_tile.hpp with the cameras of a particle SPLIT over two waves (round 6).  Device-only; included by pais_kernels.hip
// after pais_tile.hpp (whose staging -- boxes, layout, LDS-DMA -- it shares statement for statement).
//
// Why (profiles/r05_pmc_dome.txt).  k_pso_tile keeps the colours of ALL of a pixel's cameras in registers
// (mean first, then sum |c - mean|: two passes over the same colours, patch.cpp:1022-1027): 128 VGPRs of colours, 256 VGPRs
// per wave, 2 waves per SIMD -- and the walk is a chain of exposed LDS round trips (homography read -> 40 instructions ->
// byte taps -> 100 instructions) that two waves cannot cover: SQ_WAIT_ANY 47-50 % of the wave cycles with the VALU 35-49 % and
// the LDS pipe 18-31 % busy.  Eleven tunings of that kernel moved nothing.  What it needs is more waves per SIMD, i.e. fewer
// colours per wave.
//
// Mapping.  One workgroup = one candidate x 8 particles of one PSO iteration as before, but SIXTEEN waves: particle slot
// p = wave & 7 is served by wave p ("first half", role 0) and wave p + 8 ("second half", role 1).  One pixel per lane; a wave
// holds at most 2 NP colours, <= 128 VGPRs, 4 waves per SIMD (each SIMD gets two waves of either role).
//
// Arithmetic.  The kernel arithmetic sums the colours of a patch seen by >= 13 cameras in two groups (pais_eval.hpp,
// PAIS_TWO_LEVEL_K): the reference colour + the first h = 2 * ((M + 4) / 4) cameras, then the rest; likewise the absolute
// deviations.  The first half owns the first group, the second half the second (and the odd tail).  Per 64-pixel step:
//   1. both tap their cameras from the tiles and sum their group;
//   2. both post their group sum (a 512-byte LDS row + a counter), read the partner's, and compute the SAME mean
//      (sumFirst + sumSecond) / K;
//   3. both sum |c - mean| over their group; the first half posts its share and goes on to the next step;
//   4. the second half adds the two shares, weight = wStat * exp(-sad^2 / diffW), canonical sub-accumulators.
// A hand-over is a pair of LDS counters per particle (tile2_post / tile2_wait): a wave waits for ITS partner only, and only for
// its ARRIVAL -- no wave waits for arithmetic the other one has yet to do on its behalf.  (The first two versions of this kernel
// kept ONE sequential colour sum: the running sum went first -> second, the mean back, the SAD first -> second -- three hops per
// step with ~100 dependent FP64 instructions on the critical path of BOTH waves.  Correct, and 3-10 % SLOWER than k_pso_tile,
// although the same code without any hand-over -- wrong results -- ran 19.5 % faster: profiles/r06_tile2_diag.txt.  Hence the
// two-level sums.)  A patch of fewer than 13 cameras inside a many-camera batch has one group: the first half taps it all, the
// second half adds exact zeros and finishes.
// Same results as k_pso_tile and as eval_window<1, false, true, true> (tests/test_gpu_parity.py:
// test_dome_radius25_many_cameras, PAIS_TILE_VERIFY; the dome's cloud hash).
#pragma once

#define TILE2_WAVES 16
#define TILE2_SLOTS 8 // particles of a workgroup
#ifndef TILE2_LAYOUT_INTERLEAVED
#define TILE2_LAYOUT_INTERLEAVED 1
#endif
#ifndef TILE2_PAIR_MAP
#define TILE2_PAIR_MAP 0
#endif
#ifndef TILE2_SYNC
#define TILE2_SYNC 0   // 0 pair-wise LDS counters; 1 nothing (measurement build, WRONG results: what the hand-over costs)
#endif
// counter protocol, per particle slot two counters in LDS, both 0 at the start of a task; step st = 0, 1, ...:
//   first half : group sum -> rowAsum[st & 1], flagA = 2 st + 1 | waits flagB >= st + 1 | its SAD share -> rowAsad, flagA = 2 st + 2
//   second half: group sum -> rowBsum, flagB = st + 1 | waits flagA >= 2 st + 1 | ... | waits flagA >= 2 st + 2
// LDS operations of one wave are carried out in issue order, so "data, then flag" needs no more than a compiler barrier; the
// reader's data read is issued after the flag value has come back.  Re-use of the rows: rowBsum (st + 1) is written after the
// second half has finished step st, which needed flagA = 2 st + 2, posted after the first half read rowBsum (st); rowAsad (st + 1)
// is written after flagB >= st + 2, posted after the second half read rowAsad (st); rowAsum alternates by step parity -- the first
// half may write its next sum while the second half has not yet read this one.
__device__ __forceinline__ void tile2_post(volatile int *flag, int v)
{
#if TILE2_SYNC == 0
    asm volatile("" ::: "memory");
    *flag = v;
#endif
}
#ifndef TILE2_WAIT_PROFILE
#define TILE2_WAIT_PROFILE 0 // measurement builds: cycles the first-half / second-half wave of slot 0 spends waiting -> debug words 7 / 8
#endif
__device__ __forceinline__ void tile2_wait(volatile int *flag, int v)
{
#if TILE2_SYNC == 0
    while (*flag < v) {
#if TILE2_SPIN_SLEEP
        __builtin_amdgcn_s_sleep(TILE2_SPIN_SLEEP);
#endif
    }
    asm volatile("" ::: "memory");
#endif
}
#if TILE2_WAIT_PROFILE
#define TILE2_WAIT(flag, v)                                                                                         \
    {                                                                                                               \
        const unsigned long long t_ = __builtin_readcyclecounter();                                                 \
        tile2_wait(flag, v);                                                                                        \
        if (dbg && slot == 0 && lane == 0) atomicAdd(&dbg[7 + role], __builtin_readcyclecounter() - t_);            \
    }
#else
#define TILE2_WAIT(flag, v) tile2_wait(flag, v)
#endif
// the hand-over phases are short chains of DEPENDENT instructions (22 additions one after the other, the exp) on the critical
// path of a particle's two waves: at the default priority each of them queues behind the tap instructions of the SIMD's three
// other waves.  Raised priority while a wave is in such a phase, back to 0 for the taps.
#ifndef TILE2_CHAIN_PRIO
#define TILE2_CHAIN_PRIO 2
#endif
#ifndef TILE2_SPIN_SLEEP
#define TILE2_SPIN_SLEEP 3 // (0 / 1 / 3: 2 352 / 2 349 / 2 338 ms per dome reconstruction: profiles/r06_tile2_diag.txt)
#endif
__device__ __forceinline__ void tile2_prio(bool high)
{
#if TILE2_CHAIN_PRIO
    if (high) __builtin_amdgcn_s_setprio(TILE2_CHAIN_PRIO);
    else __builtin_amdgcn_s_setprio(0);
#endif
}

__host__ __device__ inline size_t tile2_fixed_lds_bytes(int Kmax)
{
    size_t b = eval_block_bytes(Kmax);                                    // EvalPatch + EvalCam[Kmax], shared by the waves
    b += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE2_SLOTS;     // homographies, per particle
    b += 2 * sizeof(TileBox) * (size_t)Kmax;                              // boxes (two sets)
    b += sizeof(double) * 64 * 4 * TILE2_SLOTS;                           // hand-over rows: 4 per particle
    b += 128;                                                             // flags[0], pstate[8], hand-over counters[8][2]
    return (b + 15) & ~(size_t)15;
}

// one camera group of the lane's pixel from the tiles: the statements of tile_tap_group<G, 1, false> without the running sum
template <int G>
__device__ __forceinline__ void tile2_tap_group(const DevScene &sc, const EvalCam *cams, const unsigned char *tiles, const double *Hbuf, int c0,
                                                double &x, double &y, double *col)
{
    asm volatile("" : "+v"(x), "+v"(y));
    double nx[G], ny[G], w[G], rw[G];
    int tbase[G], ttw[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const double2 *H2 = (const double2 *)__builtin_assume_aligned(Hbuf + PAIS_H_STRIDE * (c0 + u), 16);
        const double2 hd = H2[3], he = H2[4];
        w[u] = fma(hd.y, y, fma(hd.x, x, he.x));
        tbase[u] = __double2loint(he.y);
        ttw[u] = __builtin_amdgcn_readfirstlane(__double2hiint(he.y));
        const double2 ha = H2[0], hb = H2[1], hc = H2[2];
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
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const double ix = nx[u] * rw[u], iy = ny[u] * rw[u];
        const int px = (int)ix, py = (int)iy;
        const double bx = __builtin_amdgcn_fract(ix), by = __builtin_amdgcn_fract(iy);
        int a0, b0, a1, b1;
        if (ttw[u] != 0) { // wave-uniform: the camera's tile is staged (four BYTE reads, see tile_tap_group)
            const uint32_t a = (uint32_t)(tbase[u] + py * ttw[u] + px);
            uint32_t ar = a + 1;
            asm volatile("" : "+v"(ar));
            a0 = tiles[a]; b0 = tiles[ar];
            a1 = tiles[a + (uint32_t)ttw[u]]; b1 = tiles[ar + (uint32_t)ttw[u]];
        } else {
            TapInfo ti;
            __builtin_memcpy(&ti, __builtin_assume_aligned(&cams[c0 + u].imgOff, 16), sizeof(ti));
            const unsigned char *lvl = sc.imgBlob + ti.imgOff;
            const uint32_t off = (uint32_t)py * (uint32_t)ti.w + (uint32_t)px;
            const uint16_t r0 = load_row_at<uint16_t>(lvl, off), r1 = load_row_at<uint16_t>(lvl, off + (uint32_t)ti.w);
            a0 = r0 & 0xff; b0 = r0 >> 8;
            a1 = r1 & 0xff; b1 = r1 >> 8;
        }
        col[u] = lerp3((double)a0, (double)(b0 - a0), (double)a1, (double)(b1 - a1), bx, by);
    }
}

// corners_inside (pais_eval.hpp) for the cameras [cLo, cHi) of a particle
__device__ __forceinline__ bool corners_inside_range(const EvalPatch *ep, const EvalCam *cams, const double *Hbuf, int S, int lane, int cLo, int cHi)
{
    const int n4 = 4 * (cHi - cLo);
    bool ok = true;
    for (int t0 = 0; t0 < n4; t0 += 64) {
        const int t = t0 + lane;
        const int c = cLo + ((t < n4) ? (t >> 2) : 0), corner = t & 3;
        const double x = ep->a0 + (double)((corner & 1) ? (S - 1) : 0), y = ep->b0 + (double)((corner & 2) ? (S - 1) : 0);
        const double *H = Hbuf + PAIS_H_STRIDE * c;
        const double w = fma(H[7], y, fma(H[6], x, H[8]));
        const double rw = rcp_cr(w);
        const double ix = fma(H[1], y, fma(H[0], x, H[2])) * rw, iy = fma(H[4], y, fma(H[3], x, H[5])) * rw;
        const int qx = (int)ix, qy = (int)iy;
        const uint32_t qp = cams[c].qpack;
#if PAIS_CORNER_WTEST
        bool in = qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16) && fabs(w) > 1e-90 && fabs(w) < 1e90;
#else
        bool in = qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16);
#endif
        const unsigned long long neg = __ballot(w < 0.0), pos = __ballot(w > 0.0);
        const unsigned long long grp = 0xFull << (lane & ~3);
        in = in && (((neg & grp) == 0) || ((pos & grp) == 0)) && (((neg | pos) & grp) == grp);
        ok = ok && (in || t >= n4);
    }
    return __all(ok);
}

// Grid: candidates x particle groups of TILE2_SLOTS; workgroup of 16 waves.  Writes A.fit[i], or flags the particle pending
// (A.part[i][0] = 1) for the pending-only k_pso_eval2 launch behind it -- the interface of k_pso_tile.
template <int NP>
__global__ __launch_bounds__(64 * TILE2_WAVES) void k_pso_tile2(DevScene sc, unsigned char *states, int n, int Nmax, int Kmax,
                                                               const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win,
                                                               int tileBytes, int groups, int stripSteps, int bias, unsigned long long *dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if TILE2_PAIR_MAP == 1
    const int slot = wave >> 1, role = wave & 1;                 // (experiment: a particle's two waves on different SIMDs)
#else
    const int slot = wave & (TILE2_SLOTS - 1), role = wave >> 3; // (a SIMD's waves w, w + 4, w + 8, w + 12: two of either role)
#endif
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    size_t o = eval_block_bytes(Kmax);
    double *Hbuf = (double *)(smem + o) + (size_t)slot * Kmax * PAIS_H_STRIDE; o += sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE2_SLOTS;
    TileBox *boxAll = (TileBox *)(smem + o);                                    o += 2 * sizeof(TileBox) * (size_t)Kmax; // [2][Kmax]
    double *rowAsum = (double *)(smem + o) + (size_t)slot * 256 + lane;         // group sum of the first half: [2][64], by step parity
    double *rowBsum = rowAsum + 128;                                            // group sum of the second half
    double *rowAsad = rowAsum + 192;                                            // SAD share of the first half
    o += sizeof(double) * 64 * 4 * TILE2_SLOTS;
    int *flags = (int *)(smem + o);                                             // [0]: some particle of the group walks the tiles
    int *pstate = flags + 1;                                                    // [slot]: != 0: the particle takes the checked walk (pending)
    volatile int *flagA = flags + 1 + TILE2_SLOTS + 2 * slot, *flagB = flagA + 1;   // hand-over counters of this slot (tile2_post / tile2_wait)
    unsigned char *tiles = smem + tile2_fixed_lds_bytes(Kmax);
    const size_t SB = pso_state_bytes(Nmax);
    const int WS = win_stride(sc);
    const int S = sc.cfg.patchSize, S2 = S * S;
    const int nSteps = (S2 + 63) >> 6;
    const int nwMax = (int)(eval_block_bytes(Kmax) / 8);

    for (int task = blockIdx.x; task < n * groups; task += gridDim.x) {
        const int c = task / groups, g = task - c * groups;
        const int i = g * TILE2_SLOTS + slot; // this wave's particle
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        const int active = hd->active, Nrun = hd->N;
        if (!active || g * TILE2_SLOTS >= Nrun) continue; // uniform over the workgroup
        const bool have = i < Nrun;
        const int iLoad = have ? i : 0;
        const double theta = A.pos[iLoad][0], phi = A.pos[iLoad][1], depth = A.pos[iLoad][2];
        __syncthreads(); // the previous task's LDS is no longer read
        {
            const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
            uint64_t *dst = (uint64_t *)smem;
            for (int q = threadIdx.x; q < nwMax; q += 64 * TILE2_WAVES) dst[q] = src[q];
            for (int q = threadIdx.x; q < 2 * Kmax; q += 64 * TILE2_WAVES) boxAll[q] = TileBox{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
            if (threadIdx.x < 1 + 3 * TILE2_SLOTS) flags[threadIdx.x] = 0;
        }
        __syncthreads();
        const int M = ep->M, K = ep->K;
        const WinPix *wbase = win + (size_t)c * WS;
        // cameras 0 .. 2 nPairs - 1 in pairs, then a tail of 3, 1 or 0 (the groups of the kernel arithmetic, pais_eval.hpp);
        // the first-half wave takes pairs 0 .. pA - 1, the second-half wave the rest
        const int nPairs = (M >= 2) ? ((M & 1) ? (M - 3) / 2 : M / 2) : 0;
        const int tail0 = 2 * nPairs, nTail = M - tail0;
        // (the canonical two-level split of pais_eval.hpp: the first half owns the first group -- reference colour + hSplit cameras --,
        //  the second half the second group and the odd tail.  A patch of fewer than PAIS_TWO_LEVEL_K cameras has ONE group: the first
        //  half taps everything, the second half contributes the exact zeros 0.0 to sum and SAD and does the finishing)
        const bool twoLevel = K >= PAIS_TWO_LEVEL_K;
        const int pA = twoLevel ? two_level_split(K, M) / 2 : nPairs;
        const int tailRole = twoLevel ? 1 : 0;
        (void)bias;
        const int cLo = role ? (twoLevel ? 2 * pA : M) : 0, cHi = role ? M : (twoLevel ? 2 * pA : M); // this wave's cameras
        const int myPairs = role ? (nPairs - pA) : pA;

        // ---- the particle: normal, early exits, this wave's homographies (the statements of eval_fitness_parts)
        bool bad = false;
        if (have) {
            double nrm[3];
            wave_spherical2normal(theta, phi, nrm, lane);
            const double on[3] = {ep->optNref[0], ep->optNref[1], ep->optNref[2]};
            if (dot3(nrm, on) > 0 || !ep->valid || !(fabs(depth) > 0)) {
                bad = true;
            } else {
                double center[3];
                for (int q = 0; q < 3; ++q) center[q] = ep->ray[q] * depth + ep->Cref[q];
                const double d = -dot3(center, nrm);
                double Mref[9], invH[9], kr[9], kt[3];
                for (int q = 0; q < 9; ++q) kr[q] = ep->KRref[q];
                for (int q = 0; q < 3; ++q) kt[q] = ep->KTref[q];
                plane_matrix(d, ep->lodScale, kr, kt, nrm, Mref);
                inv3(Mref, invH);
                for (int cc = cLo + lane; cc < cHi; cc += 64) {
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
                // (more pairs in a group than this instantiation holds: cannot happen for Kmax <= 4 NP; left to the pending-only launch)
                if ((pA > NP || nPairs - pA > NP || !corners_inside_range(ep, cams, Hbuf, S, lane, cLo, cHi)) && lane == 0)
                    atomicOr(&pstate[slot], 1);
            }
        }
        __syncthreads();
        // 0: walks the tiles, 1: DBL_MAX, 2: pending (checked walk by k_pso_eval2), 3: no particle -- the same in both waves of a slot
        const int state = !have ? 3 : (bad ? 1 : (pstate[slot] ? 2 : 0));
        if (lane == 0 && have) {
            if (role == 1) {
                if (state == 1) A.fit[i] = DBL_MAX;
                A.part[i][0] = (state == 2) ? 1.0 : 0.0;
                if (dbg) atomicAdd(&dbg[state], 1ULL); // [0] particles through the tiles, [1] DBL_MAX, [2] pending
            }
            if (state == 0) flags[0] = 1;
        }
        __syncthreads();
        if (!flags[0]) continue; // nobody walks the tiles (uniform)

        const double a0 = uniform_d(ep->a0), b0 = uniform_d(ep->b0);
        const double invDiffW = uniform_d(1.0 / sc.cfg.diffWeighting);
        const bool useDiff = sc.cfg.adaptiveDifferenceEnable != 0;
        const bool hasRef = ep->hasRef != 0;
        const double invK = uniform_d(1.0 / (double)K);
        double accF[4] = {0, 0, 0, 0}, accW[4] = {0, 0, 0, 0}; // (second-half wave)

        // stage(strip): boxes -> layout -> LDS-DMA of the strip's tiles (pais_tile.hpp, the same statements; this wave's cameras
        // only in the box pass, the copy dealt to sixteen waves).  One workgroup barrier inside; every wave calls it alike.
        auto stage = [&](int s0, int par) {
            const int s1 = min(s0 + stripSteps, nSteps);
            TileBox *box = boxAll + par * Kmax;
            if (state == 0) {
                const int ya = (64 * s0) / S, yb = (min(64 * s1, S2) - 1) / S;
                for (int cc = cLo + lane; cc < cHi; cc += 64) {
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
            __syncthreads(); // the boxes are complete -- and every wave has finished the strip before this one
            int x0 = 0, y0 = 0, tw = 0, th = 0;
            if (lane < M && box[lane].xmax >= box[lane].xmin) {
                const int lw = cams[lane].w, lh = cams[lane].h;
                x0 = max(box[lane].xmin - 1, 0); y0 = max(box[lane].ymin - 1, 0);
                const int x1 = min(box[lane].xmax + 2, lw - 1), y1 = min(box[lane].ymax + 2, lh - 1);
                tw = max(((x1 - x0 + 1) + 3) & ~3, 8);
                th = y1 - y0 + 1;
            }
            const int szFull = (th > 0 && tw > (tileBytes + 1) / th) ? tileBytes + 1 : tw * th;
            const int sz = min(szFull, tileBytes + 1);
#if TILE2_LAYOUT_INTERLEAVED
            // Which cameras get a tile when the area runs out must not be one half's problem: the area is handed out in the order
            // first-group camera 0, second-group camera 0, first-group camera 1, ... (a camera that does not fit is tapped in global
            // memory -- several times slower -- and a wave waits for its partner every step).  Exclusive prefix of the sizes in that
            // order: every lane adds the sizes of the cameras that come before its own (wave-uniform broadcasts; once per strip).
            const int hA = 2 * pA; // cameras of the first group (all of them without two levels)
            const int myKey = lane < hA ? 2 * lane : 2 * (lane - hA) + 1;
            int off = 0;
            for (int j = 0; j < M; ++j) {
                const int szj = __builtin_amdgcn_readlane(sz, j);
                const int kj = j < hA ? 2 * j : 2 * (j - hA) + 1;
                off += (kj < myKey) ? szj : 0; // (sizes are saturated at tileBytes + 1 and there are <= 64 of them: no overflow)
            }
            const int incl = off + sz;
            const bool fits = sz > 0 && incl <= tileBytes;
#else
            int incl = sz;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const int up = __shfl_up(incl, m, 64);
                incl = min(incl + ((lane >= m) ? up : 0), tileBytes + 1);
            }
            const int off = incl - sz;
            const bool fits = sz > 0 && incl <= tileBytes;
#endif
            if (!fits) { tw = 0; th = 0; }
            if (lane < M) {
                // (the tile word of camera `lane` goes into this particle's record of it: written by the wave that taps it)
                if (lane >= cLo && lane < cHi) Hbuf[lane * PAIS_H_STRIDE + 9] = __hiloint2double(tw, fits ? (off - y0 * tw - x0) : 0);
                if (wave == 0) {
                    boxAll[(par ^ 1) * Kmax + lane] = TileBox{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
                    if (dbg && sz > 0) atomicAdd(&dbg[fits ? 3 : 4], 1ULL); // [3] tiles staged, [4] cameras left in global memory
                    if (dbg && fits) atomicAdd(&dbg[5], (unsigned long long)sz); // [5] bytes staged
                }
            }
            int inFlight = 0;
            for (int cc = wave; cc < M; cc += TILE2_WAVES) {
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
                        const uint32_t row = __umulhi(j, magic), dd = j - row * twd;
                        const unsigned char *gsrc = lvl + (size_t)row * lw + 4 * dd;
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
            stage(s0, sIdx & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's LDS-DMA has landed ...
            __syncthreads();                                  // ... everybody's has
            const unsigned long long tc3 = dbg ? __builtin_readcyclecounter() : 0;
            const int kpix = 64 * s0 + lane;
            int yw = kpix / S, xw = kpix - yw * S;
            const int qA = 64 / S, rA = 64 - qA * S;
            const double xs = a0 + (double)((64 * s0) % S), ys = b0 + (double)((64 * s0) / S); // the strip's first pixel
            for (int st = s0; st < s1; ++st) {
                if (state != 0) continue; // (no workgroup barrier inside a step: a particle's two waves synchronise with each other only)
                double col[2 * NP], t3[3] = {0, 0, 0};
                // ---- 1. this wave's taps of the step
                const WinPix wp = wbase[64 * st + lane]; // (the padding lanes of the last step are masked entries)
                // lanes without a pixel tap the strip's first pixel: an address inside the tiles
                const bool nopix = 64 * st + lane >= S2;
                double x = nopix ? xs : a0 + (double)xw, y = nopix ? ys : b0 + (double)yw;
                xw += rA; yw += qA;
                yw += (xw >= S) ? 1 : 0;
                xw -= (xw >= S) ? S : 0;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    if (u < myPairs) tile2_tap_group<2>(sc, cams, tiles, Hbuf, cLo + 2 * u, x, y, &col[2 * u]);
                    else col[2 * u] = col[2 * u + 1] = 0;
                }
                if (role == tailRole) {
                    if (nTail == 3) tile2_tap_group<3>(sc, cams, tiles, Hbuf, tail0, x, y, t3);
                    else if (nTail == 1) tile2_tap_group<1>(sc, cams, tiles, Hbuf, tail0, x, y, t3);
                }
                // ---- 2. this wave's group sum (pais_eval.hpp, two-level sums: the first group starts from the reference colour, the
                // second from 0), handed to the partner; the partner's; the mean -- the same operands in the same order in both waves
                tile2_prio(true);
                double sOwn = (role == 0 && hasRef) ? wp.refCol : 0.0;
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    if (u < myPairs) {
                        sOwn += col[2 * u];
                        sOwn += col[2 * u + 1];
                    }
                if (role == tailRole) { // (the tail continues the last group's sum)
                    if (nTail >= 1) sOwn += t3[0];
                    if (nTail == 3) {
                        sOwn += t3[1];
                        sOwn += t3[2];
                    }
                }
                double sA, sB;
                if (role == 0) {
                    rowAsum[(st & 1) * 64] = sOwn;
                    tile2_post(flagA, 2 * st + 1);
                    tile2_prio(false); // (a wave that spins at raised priority takes issue slots from its partner's taps)
                    TILE2_WAIT(flagB, st + 1);
                    tile2_prio(true);
                    sA = sOwn;
                    sB = *rowBsum;
                } else {
                    *rowBsum = sOwn;
                    tile2_post(flagB, st + 1);
                    tile2_prio(false);
                    TILE2_WAIT(flagA, 2 * st + 1);
                    tile2_prio(true);
                    sA = rowAsum[(st & 1) * 64];
                    sB = sOwn;
                }
                const double mean = (sA + sB) * invK;
                // ---- 3. this wave's share of the SAD; the first half hands its share over and goes on to the next step
                double sadOwn = (role == 0 && hasRef) ? fabs(wp.refCol - mean) : 0.0;
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    if (u < myPairs) {
                        sadOwn += fabs(col[2 * u] - mean);
                        sadOwn += fabs(col[2 * u + 1] - mean);
                    }
                if (role == tailRole) {
                    if (nTail >= 1) sadOwn += fabs(t3[0] - mean);
                    if (nTail == 3) {
                        sadOwn += fabs(t3[1] - mean);
                        sadOwn += fabs(t3[2] - mean);
                    }
                }
                if (role == 0) {
                    *rowAsad = sadOwn;
                    tile2_post(flagA, 2 * st + 2);
                    tile2_prio(false);
                    continue;
                }
                // ---- 4. second half: the two shares added, the weight, the canonical sub-accumulator of the step
                tile2_prio(false);
                TILE2_WAIT(flagA, 2 * st + 2);
                tile2_prio(true);
                const double sad = *rowAsad + sadOwn;
                const bool act = wp.wStat >= 0.0;
                const double sadq = sad * invK;
                double weight = wp.wStat;
                if (useDiff) weight *= det_exp_poly(mul_uniform(-(sadq * sadq), invDiffW));
                const int ga = st & 3; // (uniform)
#define PAIS_TACC(a)                                          \
    {                                                         \
        accW[a] = act ? (accW[a] + weight) : accW[a];         \
        accF[a] = act ? fma(weight, sadq, accF[a]) : accF[a]; \
    }
                if (ga == 0) PAIS_TACC(0) else if (ga == 1) PAIS_TACC(1) else if (ga == 2) PAIS_TACC(2) else PAIS_TACC(3)
#undef PAIS_TACC
                tile2_prio(false);
            }
            if (dbg) tWalk += __builtin_readcyclecounter() - tc3;
        }
        if (dbg && threadIdx.x == 0) {
            atomicAdd(&dbg[6], __builtin_readcyclecounter() - tc0 - tWalk); // staging (boxes, layout, DMA issue, barriers) ...
            atomicAdd(&dbg[9], tWalk);                                      // ... and the walks of wave 0
        }
        if (state == 0 && role == 1) {
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
