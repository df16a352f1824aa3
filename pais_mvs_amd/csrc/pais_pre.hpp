// pais_pre.hpp -- the per-evaluation set-up of PAIS::getFitness for a WHOLE SWARM by one wave (round 6).
// Device-only; included by pais_kernels.hip behind the coherent load / store helpers.
//
// Before its window walk an evaluation turns the particle (theta, phi, depth) into the plane's normal (utility.h:25-29), the
// early exits of patch.cpp:939-962, the plane-induced homographies of the M other cameras (patch.cpp:290-330) and the
// corner test that selects the unchecked walk (pais_eval.hpp corners_inside).  Inside an evaluation wave that is ~570 VALU
// instructions of wave-uniform or nearly wave-uniform work per evaluation -- a fifth of an evaluation at five cameras
// (profiles/r06_pre_setup_ab.txt) -- because a wave instruction costs its issue slot whether 1 or 64 lanes have something
// to do.  The swarm step knows all N new positions at once, so it does this work with the particles ACROSS the lanes:
//   stage 0   lane = (particle, one of sin theta / cos theta / sin phi / cos phi): the four fdlibm evaluations of every
//             particle in one pass (wave_spherical2normal's statement, one function value per lane)
//   stage 1   lane = particle: normal, early exits, plane matrix of the reference camera and its inverse
//   stage 2   lane = (particle, camera): homography, tap word, the four window corners
//   stage 3   lane = particle: status word
// and the evaluation wave of (candidate, particle) reads a record {status, H[M][10]} instead of computing it.  Every value is
// produced by the same operations on the same operands as in eval_fitness_parts (the compiler contracts nothing:
// -ffp-contract=off) -- the records, fitness values and clouds are the same bits (tests: pipeline shapes, ring pass, goldens).
#pragma once

#define PAIS_PRE_HDR 2 // doubles in front of a record's homographies: [0] status (0: unchecked walk, 1: the call is DBL_MAX, 2: checked walk)
#define PAIS_PRE_SCR 18 // doubles of LDS scratch per particle: 4 sin / cos values, normal, d, inverse reference plane matrix, (spare)
__host__ __device__ inline size_t pre_rec_doubles(int Kmax) { return PAIS_PRE_HDR + PAIS_H_STRIDE * (size_t)Kmax; }
__host__ __device__ inline size_t pre_rec_bytes(int Kmax) { return sizeof(double) * pre_rec_doubles(Kmax); }
// LDS scratch of swarm_eval_setup for a swarm of N particles (bytes; 16-byte multiple)
__host__ __device__ inline size_t pre_scratch_bytes(int N) { return (sizeof(double) * PAIS_PRE_SCR * (size_t)N + sizeof(int) * (size_t)N + 15) & ~(size_t)15; }

// ep / cams: the candidate's evaluation block in LDS; pos: the swarm's positions in LDS; scr: pre_scratch_bytes(N) of LDS;
// rec: the candidate's N records in global memory (stride recD doubles).  COH: stored for other waves of THIS launch (k_pso_ring).
template <bool COH>
__device__ void swarm_eval_setup(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, const double (*pos)[3], int N, double *scr,
                                 double *rec, size_t recD, int lane)
{
    int *statusI = (int *)(scr + (size_t)PAIS_PRE_SCR * N);
    const int M = ep->M;
    const int S = sc.cfg.patchSize;
    // stage 0: sin(theta), cos(theta), sin(phi), cos(phi) -- one per lane (the statements of wave_spherical2normal)
    for (int t = lane; t < 4 * N; t += 64) {
        const int p = t >> 2, which = t & 3;
        const double arg = (which & 2) ? pos[p][1] : pos[p][0];
        const int wantCos = which & 1;
        double y0, y1;
        const int q = (det_rem_pio2(arg, &y0, &y1) + wantCos) & 3;
        const double ks = det_ksin(y0, y1, 1), kc = det_kcos(y0, y1);
        double r = (q & 1) ? kc : ks;
        r = (q & 2) ? -r : r;
        if (arg != arg || arg - arg != 0.0) r = arg - arg; // NaN / inf, as det_sin / det_cos
        scr[PAIS_PRE_SCR * p + which] = r;
    }
    wave_sync();
    // stage 1: normal, the early exits (patch.cpp:939, :952-962), the reference camera's plane matrix and its inverse (:314)
    for (int p = lane; p < N; p += 64) {
        double *my = scr + PAIS_PRE_SCR * p;
        const double st = my[0], ct = my[1], sp = my[2], cp = my[3];
        double n[3];
        n[0] = st * cp;
        n[1] = st * sp;
        n[2] = ct;
        const double depth = pos[p][2];
        int status = 0;
        {
            double on[3] = {ep->optNref[0], ep->optNref[1], ep->optNref[2]};
            if (dot3(n, on) > 0) status = 1;
        }
        if (!ep->valid) status = 1;
        if (!(fabs(depth) > 0)) status = 1;
        double center[3];
        for (int i = 0; i < 3; ++i) center[i] = ep->ray[i] * depth + ep->Cref[i]; // :944
        const double d = -dot3(center, n);
        double Mref[9], invH[9], kr[9], kt[3];
        for (int i = 0; i < 9; ++i) kr[i] = ep->KRref[i];
        for (int i = 0; i < 3; ++i) kt[i] = ep->KTref[i];
        plane_matrix(d, ep->lodScale, kr, kt, n, Mref);
        inv3(Mref, invH);
        my[4] = n[0];
        my[5] = n[1];
        my[6] = n[2];
        my[7] = d;
        for (int i = 0; i < 9; ++i) my[8 + i] = invH[i];
        statusI[p] = status;
    }
    wave_sync();
    // stage 2: homography of (particle, camera), tap word, the corner test of the unchecked walk (corners_inside, four corners in turn)
    const int pairs = N * M;
    for (int t = lane; t < pairs; t += 64) {
        const int p = t / M, c = t - p * M;
        const double *my = scr + PAIS_PRE_SCR * p;
        double H[9];
        if (cams[c].cam == ep->refCam) { // :317-320 (a second occurrence of the reference camera)
            H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
        } else {
            double n[3] = {my[4], my[5], my[6]}, invH[9], kr[9], kt[3], Mc[9];
            for (int i = 0; i < 9; ++i) invH[i] = my[8 + i];
            for (int i = 0; i < 9; ++i) kr[i] = cams[c].KR[i];
            for (int i = 0; i < 3; ++i) kt[i] = cams[c].KT[i];
            plane_matrix(my[7], ep->lodScale, kr, kt, n, Mc);
            mul33(Mc, invH, H);
        }
        double *out = rec + recD * (size_t)p + PAIS_PRE_HDR + PAIS_H_STRIDE * c;
        for (int i = 0; i < 9; ++i) sstore<COH>(&out[i], H[i]);
        sstore<COH>(&out[9], __longlong_as_double((long long)((cams[c].imgOff & 0xFFFFFFFFFFull) | ((uint64_t)(uint32_t)cams[c].w << 40))));
        const uint32_t qp = cams[c].qpack;
        bool in = true, allNeg = true, allPos = true;
#pragma unroll 1
        for (int corner = 0; corner < 4; ++corner) {
            const double x = ep->a0 + (double)((corner & 1) ? (S - 1) : 0), y = ep->b0 + (double)((corner & 2) ? (S - 1) : 0);
            const double w = fma(H[7], y, fma(H[6], x, H[8]));
            const double rw = rcp_cr(w);
            const double ix = fma(H[1], y, fma(H[0], x, H[2])) * rw, iy = fma(H[4], y, fma(H[3], x, H[5])) * rw;
            const int qx = (int)ix, qy = (int)iy;
#if PAIS_CORNER_WTEST
            in = in && qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16) && fabs(w) > 1e-90 && fabs(w) < 1e90;
#else
            in = in && qx >= 3 && qx < (int)(qp & 0xffffu) && qy >= 3 && qy < (int)(qp >> 16);
#endif
            allNeg = allNeg && (w < 0.0);
            allPos = allPos && (w > 0.0);
        }
        if (!(in && (allNeg || allPos)) || !PAIS_CORNER_FASTPATH) atomicOr(&statusI[p], 2);
    }
    wave_sync();
    for (int p = lane; p < N; p += 64) {
        const int s = statusI[p];
        sstore<COH>(&rec[recD * (size_t)p], __longlong_as_double((long long)((s & 1) ? 1 : ((s & 2) ? 2 : 0))));
    }
}

// the evaluation of one particle from its record (what eval_fitness_parts does behind its set-up); hv: the lane's first word of
// the record's homographies, requested by the caller together with its other loads
template <int NS, bool BYTES, bool ACCR, bool COH>
__device__ int eval_fitness_pre(const DevScene &sc, const EvalPatch *ep, const EvalCam *cams, double *Hbuf, double *cbuf, const WinPix *win,
                                const double *rec, double status0, double hv, int lane, double *f4, double *w4)
{
    f4[0] = f4[1] = f4[2] = f4[3] = 0;
    w4[0] = w4[1] = w4[2] = w4[3] = 0;
    const int status = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(status0));
    if (status == 1) return 1;
    const int nH = PAIS_H_STRIDE * ep->M;
    if (lane < nH) Hbuf[lane] = hv;
    for (int j = lane + 64; j < nH; j += 64) Hbuf[j] = sload<COH>(&rec[PAIS_PRE_HDR + j]);
    wave_sync();
    if (status == 0) return eval_window<NS, false, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win, lane, 0, 1, f4, w4);
    return eval_window<NS, true, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win, lane, 0, 1, f4, w4);
}
