// pais_literal.hpp -- PAIS::getFitness (TMVS/mvs/patch.cpp:914-1047) in the REFERENCE'S OWN arithmetic: its per-pixel
// expressions and its summation order.  Selected with PAIS_ARITH=literal (round 6); device-only, included by pais_kernels.hip.
//
// The default evaluation (pais_eval.hpp, "kernel arithmetic") computes the same real-number function re-associated for a
// wave: the window origin once per run, fused multiply-adds in the homography rows, one reciprocal per camera pair,
// lerp-form bilinear taps, 1/K instead of /K, a polynomial exp, lane partial sums + butterflies.  Its values differ from the
// reference's in the last bits, and on a scene whose PSO runs do not converge a last-bit difference decides a
// `fitness < pBestFitness` the other way.  This file is the other end of that trade: one wave per
// evaluation as before, but
//   * the window from the particle's own centre (patch.cpp:944-962), x and y by the reference's repeated ++ (:979-980);
//   * per pixel and camera, in camIdx order (the reference camera at its own position, through its identity homography):
//     w = h6 x + h7 y + h8 without contraction, ix = (h0 x + h1 y + h2) / w with a true division (:994-996), the bounds
//     test of :999 on the doubles, the four-product bilinear of :1014-1017;
//   * mean /= K, avgSad /= K (:1022-1027), weight = 1 * dist * exp(-avgSad * avgSad / diffW) * exp(-1 / (edge * gradW)) in
//     that order (:1029-1038), with fdlibm's exp (det_exp: the platform libm of the reference's MSVC build is not
//     reproducible anywhere else; the CPU checker of the tests uses the same function);
//   * sumWeight += weight; fitness += weight * avgSad over the pixels in the reference's x-outer / y-inner order (:979-1041):
//     a wave computes 64 consecutive pixels of that order at a time and every lane adds the 64 (weight, weight * avgSad) one after
//     the other (v_readlane broadcasts) -- a strict sequential sum, bit for bit the CPU's.
// Checker: the CPU restatement's literal cost with the same exp / sin / cos (tests/test_gpu_parity.py:
// test_literal_arithmetic_cost_is_the_reference_statement).  Slower than the kernel arithmetic (a division per tap, two
// serial chains of S*S additions per evaluation); bench.py reports its throughput next to the default's.
#pragma once
#ifndef PAIS_LIT_SUM_LDS
#define PAIS_LIT_SUM_LDS 0
#endif

// LDS of one evaluating wave: [EvalPatch][EvalCam x Kmax] [H: Kmax x 9] [colour rows: Kmax x 64] [xs, ys: 2 x 64] [w, wf: 2 x 64]
__host__ __device__ inline size_t literal_lds_bytes(int Kmax)
{
    return eval_block_bytes(Kmax) + sizeof(double) * (9 * (size_t)Kmax + 64 * (size_t)Kmax + 4 * 64);
}

// returns the cost of particle (theta, phi, depth); every lane gets the same value
__device__ double eval_fitness_literal(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *crow, double *xs,
                                       double *ys, double *srow, double theta, double phi, double depth, int lane)
{
    double n[3];
    wave_spherical2normal(theta, phi, n, lane); // (the bits of det_sin / det_cos: utility.h:25-29 with the deterministic libm)
    {
        const double on[3] = {ep->optNref[0], ep->optNref[1], ep->optNref[2]};
        if (dot3(n, on) > 0) return DBL_MAX; // patch.cpp:939
    }
    const int K = ep->K, M = ep->M, LOD = ep->LOD, refCam = ep->refCam;
    const int refPos = ep->hasRef ? ep->refPos : -1; // position of the reference camera in camIdx (its first occurrence)
    const DevCamera &rc = sc.cams[refCam];
    double center[3];
    for (int i = 0; i < 3; ++i) center[i] = ep->ray[i] * depth + ep->Cref[i]; // :944
    // homographies of the other cameras (:290-330; the statements of eval_fitness_parts)
    {
        const double d = -dot3(center, n);
        double Mref[9], invH[9], kr[9], kt[3];
        for (int i = 0; i < 9; ++i) kr[i] = ep->KRref[i];
        for (int i = 0; i < 3; ++i) kt[i] = ep->KTref[i];
        plane_matrix(d, ep->lodScale, kr, kt, n, Mref);
        inv3(Mref, invH);
        for (int c = lane; c < M; c += 64) {
            double H[9];
            if (cams[c].cam == refCam) { // :317-320 (a second occurrence of the reference camera)
                H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
            } else {
                double Mc[9];
                for (int i = 0; i < 9; ++i) kr[i] = cams[c].KR[i];
                for (int i = 0; i < 3; ++i) kt[i] = cams[c].KT[i];
                plane_matrix(d, ep->lodScale, kr, kt, n, Mc);
                mul33(Mc, invH, H);
            }
            for (int i = 0; i < 9; ++i) Hbuf[c * 9 + i] = H[i];
        }
    }
    // the particle's own window (:952-962)
    const int refW = rc.w[LOD], refH = rc.h[LOD];
    const int r = sc.cfg.patchRadius, S = sc.cfg.patchSize;
    double pt[2];
    project_raw(rc.R, rc.T, rc.focal, rc.pp, ep->lodScale, center, pt);
    if (!(LOD <= rc.maxLOD && in_image_d(pt, refW, refH))) return DBL_MAX;                                     // :952
    if (pt[0] - r < 2 || pt[0] + r >= refW - 3 || pt[1] - r < 2 || pt[1] + r >= refH - 3) return DBL_MAX;       // :957-962
    // x and y of the walk: start + 1 + 1 + ... (:979-980), by one lane; at most S values of either (the distance table has S x S)
    int nx = 0, ny = 0;
    {
        wave_sync();
        if (lane == 0) {
            int k = 0;
            for (double x = pt[0] - r; x <= pt[0] + r && k < S && k < 64; ++x) xs[k++] = x;
            srow[0] = (double)k;
        } else if (lane == 1) {
            int k = 0;
            for (double y = pt[1] - r; y <= pt[1] + r && k < S && k < 64; ++y) ys[k++] = y;
            srow[1] = (double)k;
        }
        wave_sync();
        nx = (int)srow[0];
        ny = (int)srow[1];
    }
    const uint8_t *refImg = sc.imgBlob + rc.imgOff[LOD];
    const double *refEdge = sc.edgeBlob ? sc.edgeBlob + rc.edgeOff[LOD] : nullptr;
    const double eMin = rc.edgeMin[LOD], eMax = rc.edgeMax[LOD];
    const bool useDist = sc.cfg.adaptiveDistanceEnable != 0, useDiff = sc.cfg.adaptiveDifferenceEnable != 0,
               useGrad = sc.cfg.adaptiveGradientEnable != 0;
    const double diffW = sc.cfg.diffWeighting, gradW = sc.cfg.gradientWeighting;
    double *myc = crow + lane;
    double fitness = 0, sumWeight = 0;
    const int total = nx * ny;
    for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        const bool have = k < total;
        const int kk = have ? k : 0;
        const int xi = kk / ny, yi = kk - xi * ny;
        const double x = xs[xi], y = ys[yi];
        const int rx = cv_round(x), ry = cv_round(y);
        const bool live = have && refImg[(size_t)ry * refW + rx] != 0; // :986 (a masked pixel is skipped before any tap)
        bool over = false;
        double mean = 0;
        int slot = 0;
        for (int i = 0; i < K; ++i) {
            const bool isRef = (i == refPos); // wave-uniform
            const uint8_t *img;
            int cols, rows;
            double w, ix, iy;
            if (isRef) { // the identity homography (:317-320) through the same expressions
                img = refImg; cols = refW; rows = refH;
                w = (0.0 * x + 0.0 * y + 1.0);
                ix = (1.0 * x + 0.0 * y + 0.0) / w;
                iy = (0.0 * x + 1.0 * y + 0.0) / w;
            } else {
                const double *Hi = Hbuf + 9 * slot;
                img = sc.imgBlob + cams[slot].imgOff; cols = cams[slot].w; rows = cams[slot].h;
                w = (Hi[6] * x + Hi[7] * y + Hi[8]);           // :994
                ix = (Hi[0] * x + Hi[1] * y + Hi[2]) / w;      // :995
                iy = (Hi[3] * x + Hi[4] * y + Hi[5]) / w;      // :996
                ++slot;
            }
            const bool bad = ix < 2 || ix >= cols - 3 || iy < 2 || iy >= rows - 3 || w == 0 || ix != ix || iy != iy; // :999
            over = over || bad;
            const bool tap = live && !bad;
            const int px0 = tap ? (int)ix : 2, py0 = tap ? (int)iy : 2; // (an address inside the level for the lanes that do not count)
            const int px1 = px0 + 1, py2 = py0 + 1;
            const uint8_t *r0 = img + (size_t)py0 * cols + px0, *r1 = r0 + cols;
            const double c = (double)r0[0] * (px1 - ix) * (py2 - iy) + (double)r0[1] * (ix - px0) * (py2 - iy) +
                             (double)r1[0] * (px1 - ix) * (iy - py0) + (double)r1[1] * (ix - px0) * (iy - py0); // :1014-1017
            myc[i * 64] = c;
            mean += c;
        }
        if (__any(live && over)) return DBL_MAX; // :1001 -- whole call
        mean /= K;                                // :1022
        double avgSad = 0;
        for (int i = 0; i < K; ++i) avgSad += fabs(myc[i * 64] - mean);
        avgSad /= K;                              // :1027
        double weight = 1;
        if (useDist) weight *= sc.gauss[kk];      // :1031 (the table iterator advances once per pixel of the walk: index k)
        if (useDiff) weight *= det_exp(-avgSad * avgSad / diffW); // :1034
        if (useGrad) {
            const double e = refEdge ? refEdge[(size_t)ry * refW + rx] : edge_on_the_fly(refImg, refW, refH, rx, ry, eMin, eMax);
            weight *= det_exp(-1.0 / (e * gradW)); // :1037
        }
        // the 64 pixels of this trip, added in the reference's order by every lane alike: lane j's (weight, weight * avgSad) are
        // broadcast with v_readlane (an SGPR lane index: no LDS round trip inside the two serial chains of additions)
        const double wf = weight * avgSad;
        unsigned long long todo = __ballot(live);
#if PAIS_LIT_SUM_LDS // (A/B build: the 64 values parked in LDS and read back with wave-uniform reads)
        wave_sync();
        srow[lane] = weight;
        srow[64 + lane] = wf;
        wave_sync();
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            sumWeight += srow[j];
            fitness += srow[64 + j];
        }
#else
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            sumWeight += lane_get(weight, j); // :1040
            fitness += lane_get(wf, j);       // :1041
        }
#endif
    }
    return fitness / sumWeight; // :1046 (NaN when every pixel was masked)
}

// pais_fitness_batch in the literal arithmetic: one wave (= one workgroup) per evaluation
__global__ __launch_bounds__(64) void k_fitness_lit(DevScene sc, const int32_t *stateIndex, const double *particles, double *out, int nEvals,
                                                    int Kmax, const unsigned char *evalBlocks, size_t evalBlockBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *crow = Hbuf + 9 * (size_t)Kmax, *xs = crow + 64 * (size_t)Kmax, *ys = xs + 64, *srow = ys + 64;
    const int lane = threadIdx.x;
    const int nw = (int)(eval_block_bytes(Kmax) / 8);
    for (int e = blockIdx.x; e < nEvals; e += gridDim.x) {
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)stateIndex[e]);
        wave_sync();
        for (int q = lane; q < nw; q += 64) ((uint64_t *)smem)[q] = src[q];
        wave_sync();
        const double v = eval_fitness_literal(sc, ep, cams, Hbuf, crow, xs, ys, srow, particles[3 * e], particles[3 * e + 1], particles[3 * e + 2], lane);
        if (lane == 0) out[e] = v;
    }
}

// the evaluation launch of a PSO iteration in the literal arithmetic: one wave per (candidate, particle) -- the interface of
// k_pso_eval2; k_pso_step follows it as in the large-batch pipeline
__global__ __launch_bounds__(64) void k_pso_eval_lit(DevScene sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks,
                                                     size_t evalBlockBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *crow = Hbuf + 9 * (size_t)Kmax, *xs = crow + 64 * (size_t)Kmax, *ys = xs + 64, *srow = ys + 64;
    const int lane = threadIdx.x;
    const size_t SB = pso_state_bytes(Nmax);
    const int total = n * Nmax;
    const int nw = (int)(eval_block_bytes(Kmax) / 8);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int c = t / Nmax, i = t - c * Nmax;
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        if (!hd->active || i >= hd->N) continue;
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
        wave_sync();
        for (int q = lane; q < nw; q += 64) ((uint64_t *)smem)[q] = src[q];
        wave_sync();
        const double v = eval_fitness_literal(sc, ep, cams, Hbuf, crow, xs, ys, srow, A.pos[i][0], A.pos[i][1], A.pos[i][2], lane);
        if (lane == 0) A.fit[i] = v;
    }
}
