// pais_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the PAIS-MVS hot path.
//
//   k_begin      : head of Patch::refine()                         (patch.cpp:114-136)
//   k_pso_init   : psoOptimization set-up, initial swarm, the candidate's evaluation constants   (patch.cpp:180-200,
//                                                                                                  psosolver.cpp:94-110)
//   k_pso_eval2  : PAIS::getFitness for one (candidate, particle) per wave          (patch.cpp:914-1047)   } large
//   k_pso_step   : PsoSolver::run() between two fitness passes, one wave per candidate   (psosolver.cpp)    } batches
//   k_pso_iter   : the same step replayed inside the evaluation waves + getFitness: one launch per iteration (small
//                  batches; 1 / 2 / 4 waves per evaluation)
//   k_after      : removeInvisibleCamera + setters + seed-loop control  (patch.cpp:156-175, 655-721; mvs.cpp:215,574)
//   k_fitness    : batched PAIS::getFitness (pais_fitness_batch)
//   k_neighbor_count : all-pairs neighbour counts of MVS::neighborPatchFiltering   (mvs.cpp:448-524)
//
// Mapping: one wave per cost evaluation, window pixels <-> lanes (row-major, so neighbouring
// lanes read neighbouring pixels of the reprojected image rows), the swarm of a candidate <-> lanes in the step, FP64
// xor-butterfly reductions in a fixed order so that a cost value is a pure function of (patch, particle).  Homographies,
// camera constants and the per-camera colours of a pixel live in LDS.  No MFMA: the work is gathers + FP64 VALU
// (SURVEY 8d).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/pais_hip.h"
#include "pais_dev.hpp"
#include "pais_internal.h"

using namespace pais;

// Compile-time tuning of the cost evaluation (scripts/sens_variants.py builds variants):
//   PAIS_RCP_NEWTON  1: reciprocal of a camera group as estimate + Newton + Markstein correction (pais_eval.hpp rcp_cr)
#ifndef PAIS_RCP_NEWTON
#define PAIS_RCP_NEWTON 1
#endif
//   PAIS_ACC_REG     1: the lane's four (fitness, weight) sub-accumulators live in registers instead of LDS rows
#ifndef PAIS_ACC_REG
#define PAIS_ACC_REG 0
#endif
//   PAIS_CORNER_FASTPATH  1: evaluations whose window corners map inside every image skip the per-tap bounds logic
#ifndef PAIS_CORNER_FASTPATH
#define PAIS_CORNER_FASTPATH 1
#endif
//   PAIS_WAVE_SINCOS  1: the normal of a particle by four lanes in one pass (pais_eval.hpp wave_spherical2normal)
#ifndef PAIS_WAVE_SINCOS
#define PAIS_WAVE_SINCOS 1
#endif
//   PAIS_CORNER_WTEST 1: corners_inside also bounds the size of the denominators
#ifndef PAIS_CORNER_WTEST
#define PAIS_CORNER_WTEST 1
#endif
//   PAIS_WG_WAVES    waves (= consecutive evaluation tasks: particles of one candidate) per workgroup of the evaluation
//                    kernels: the waves of a workgroup run on one CU and share its L1 -- the taps of a candidate's particles
//                    fall into the same few image windows.  LDS scratch stays private to each wave: no barriers.
#ifndef PAIS_WG_WAVES
#define PAIS_WG_WAVES 1
#endif
//   PAIS_XCD_SWIZZLE 1: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md); remap so that consecutive tasks share an XCD's L2
#ifndef PAIS_XCD_SWIZZLE
#define PAIS_XCD_SWIZZLE 0
#endif
//   PAIS_NS1_WAVES   waves per SIMD the register allocator is asked for in the one-pixel-per-lane kernels (many cameras)
#ifndef PAIS_NS1_WAVES
#define PAIS_NS1_WAVES 2   // only batches of more than 12 cameras run them: their LDS scratch allows < 2 waves per SIMD anyway
#endif
#ifndef PAIS_NS2_WAVES
#define PAIS_NS2_WAVES 3
#endif
#define PAIS_EVAL_BOUNDS(NS) __launch_bounds__(64 * PAIS_WG_WAVES, (NS) == 1 ? PAIS_NS1_WAVES : PAIS_NS2_WAVES)
// k_pso_iter launches that share an evaluation among several waves run with the GPU nearly empty and are bound by the latency
// of one wave: no occupancy to protect, so the register allocator gets the whole file (no scratch reloads -- each a memory
// round trip -- on the step replay's critical path)
#ifndef PAIS_ITER_FREE_REGS_FROM
#define PAIS_ITER_FREE_REGS_FROM 2
#endif
#define PAIS_ITER_BOUNDS(P, NS) __launch_bounds__(64 * PAIS_WG_WAVES, (P) >= PAIS_ITER_FREE_REGS_FROM ? 1 : ((NS) == 1 ? PAIS_NS1_WAVES : PAIS_NS2_WAVES))
//   PAIS_TWO_PIXELS_MAXK  largest camera count of a batch that still runs two window pixels per lane (NS = 2)
#ifndef PAIS_TWO_PIXELS_MAXK
#define PAIS_TWO_PIXELS_MAXK 6
#endif

// --------------------------------------------------------------- helpers ---
// first task / task stride of this wave in the evaluation kernels (grid-stride over tasks; PAIS_WG_WAVES tasks per workgroup)
__device__ __forceinline__ int eval_first_task()
{
    int b = (int)blockIdx.x;
#if PAIS_XCD_SWIZZLE
    const int per = (int)gridDim.x >> 3; // the launchers make the grid a multiple of 8
    b = (b & 7) * per + (b >> 3);
#endif
    return b * PAIS_WG_WAVES + (int)(threadIdx.x >> 6);
}
__device__ __forceinline__ int eval_task_stride() { return (int)gridDim.x * PAIS_WG_WAVES; }

__device__ __forceinline__ void wave_sync()
{
    // LDS/global traffic of one wave is processed in issue order; a
    // wavefront-scope fence only has to stop the compiler from reordering.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// value of lane `src` for a WAVE-UNIFORM src (a loop counter, this wave's particle, the winner of a wave-wide selection):
// two v_readlane_b32 through an SGPR index instead of two ds_bpermute_b32 through the LDS crossbar -- the step replay of
// k_pso_iter is chains of such broadcasts (one per particle and quantity), each a ~130-cycle round trip as a bpermute
__device__ __forceinline__ double lane_get(double v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), s), hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_x(double v);
// the wave64 xor butterfly of every reduction of the path (m = 32, 16, ..., 1): wave_sum_x below computes it without the LDS crossbar
// (round 4: six __shfl_xor = twelve ds_bpermute_b32 per sum)
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_x(v); }
// The same xor butterfly (m = 32, 16, 8, 4, 2, 1: at every level a lane adds its partner's value, both partners get the same sum)
// without the LDS crossbar -- round 5: the eight reductions of an evaluation were 96 ds_bpermute_b32, a quarter of its LDS
// instructions and six dependent ~130-cycle round trips each.  gfx950: v_permlane32_swap / v_permlane16_swap exchange halves /
// rows between two registers (given v twice they leave {v[l], v[l ^ m]} in some order in every lane: the sum is the same either
// way, IEEE addition commutes); xor 8 is a row rotation by 8, xor 4 a half-row mirror followed by a quad reversal, xor 2 and 1
// quad permutations: DPP operand modifiers of v_mov.  Same pairs, same sums, same bits.  ALL 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ double wsum_dpp(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double wave_sum_x(double v)
{
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
    }
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
    }
    v += wsum_dpp<0x128>(v);                  // row_ror:8          l ^ 8
    v += wsum_dpp<0x1B>(wsum_dpp<0x141>(v));  // row_half_mirror, then quad_perm [3, 2, 1, 0]: l ^ 4
    v += wsum_dpp<0x4E>(v);                   // quad_perm [2, 3, 0, 1]  l ^ 2
    v += wsum_dpp<0xB1>(v);                   // quad_perm [1, 0, 3, 2]  l ^ 1
    return v;
}
// EIGHT butterflies at once (round 6): the (fitness, weight) pairs of an evaluation's four canonical sub-accumulators.  As eight
// wave_sum_x they were ~200 VALU instructions at the end of every evaluation (80 DPP moves, 32 permlane swaps, 48 additions, the
// moves around them) -- 6 % of an evaluation at five cameras.  A level of the butterfly leaves the same sum in BOTH partners, so one
// of them is free to carry another value: v_permlane32_swap(x, y) puts {x[l], x[l + 32]} into the lower 32 lanes and {y[l - 32], y[l]}
// into the upper 32 -- ONE addition then holds the m = 32 level of x below and of y above; v_permlane16_swap does the same for the
// 16-lane rows.  Two levels turn eight values into two registers (four additions + two instead of sixteen); the levels m = 8 ... 1 run
// as before on those two; the eight totals are read from lanes 0 / 16 / 32 / 48.  The SAME pairs are added at every level of every
// value (IEEE addition commutes): the bits of wave_sum_x.  ALL 64 lanes must be active.
__device__ __forceinline__ double wsum_swapadd32(double x, double y)
{
    const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ double wsum_swapadd16(double x, double y)
{
    const auto rl = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ double wsum_const_lane(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// a[0 .. 7] per lane -> t[j] = wave_sum_x(a[j]), wave-uniform
__device__ __forceinline__ void wave_sum8(const double *a, double *t)
{
    // m = 32: lanes 0..31 hold the level of a[j], lanes 32..63 that of a[j + 4]
    const double z0 = wsum_swapadd32(a[0], a[4]), z1 = wsum_swapadd32(a[1], a[5]);
    const double z2 = wsum_swapadd32(a[2], a[6]), z3 = wsum_swapadd32(a[3], a[7]);
    // m = 16: rows 0 / 1 / 2 / 3 of w0 hold a[0] / a[2] / a[4] / a[6], of w1 a[1] / a[3] / a[5] / a[7]
    double w0 = wsum_swapadd16(z0, z2), w1 = wsum_swapadd16(z1, z3);
    w0 += wsum_dpp<0x128>(w0);                  w1 += wsum_dpp<0x128>(w1);                  // l ^ 8
    w0 += wsum_dpp<0x1B>(wsum_dpp<0x141>(w0));  w1 += wsum_dpp<0x1B>(wsum_dpp<0x141>(w1));  // l ^ 4
    w0 += wsum_dpp<0x4E>(w0);                   w1 += wsum_dpp<0x4E>(w1);                   // l ^ 2
    w0 += wsum_dpp<0xB1>(w0);                   w1 += wsum_dpp<0xB1>(w1);                   // l ^ 1
    t[0] = wsum_const_lane(w0, 0);  t[2] = wsum_const_lane(w0, 16); t[4] = wsum_const_lane(w0, 32); t[6] = wsum_const_lane(w0, 48);
    t[1] = wsum_const_lane(w1, 0);  t[3] = wsum_const_lane(w1, 16); t[5] = wsum_const_lane(w1, 32); t[7] = wsum_const_lane(w1, 48);
}
// the same for FOUR values (an evaluation shared by two waves: two sub-accumulators each) and for TWO (four waves)
__device__ __forceinline__ void wave_sum4(const double *a, double *t)
{
    const double z0 = wsum_swapadd32(a[0], a[2]), z1 = wsum_swapadd32(a[1], a[3]); // lanes 0..31: a[0] / a[1], lanes 32..63: a[2] / a[3]
    double w = wsum_swapadd16(z0, z1);                                              // rows 0 / 1 / 2 / 3: a[0] / a[1] / a[2] / a[3]
    w += wsum_dpp<0x128>(w);
    w += wsum_dpp<0x1B>(wsum_dpp<0x141>(w));
    w += wsum_dpp<0x4E>(w);
    w += wsum_dpp<0xB1>(w);
    t[0] = wsum_const_lane(w, 0); t[1] = wsum_const_lane(w, 16); t[2] = wsum_const_lane(w, 32); t[3] = wsum_const_lane(w, 48);
}
__device__ __forceinline__ void wave_sum2(const double *a, double *t)
{
    double w = wsum_swapadd32(a[0], a[1]); // lanes 0..31: a[0], lanes 32..63: a[1]
    w = wsum_swapadd16(w, w);
    w += wsum_dpp<0x128>(w);
    w += wsum_dpp<0x1B>(wsum_dpp<0x141>(w));
    w += wsum_dpp<0x4E>(w);
    w += wsum_dpp<0xB1>(w);
    t[0] = wsum_const_lane(w, 0); t[1] = wsum_const_lane(w, 32);
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

#include "pais_eval.hpp"

// ------------------------------------------------------------- k_fitness ---
// pais_fitness_batch: k_state_blocks builds the evaluation block of every patch state (one wave per state), then
// one wave (= one 64-thread workgroup) per evaluation
__global__ __launch_bounds__(64) void k_state_blocks(DevScene sc, const pais_patch_state *states, int nStates, unsigned char *evalBlocks,
                                                     size_t evalBlockBytes, WinPix *win)
{
    const int lane = threadIdx.x;
    const int WS = win_stride(sc);
    for (int s = blockIdx.x; s < nStates; s += gridDim.x) {
        const pais_patch_state *st = &states[s];
        unsigned char *blk = evalBlocks + evalBlockBytes * (size_t)s;
        build_eval_block(sc, (EvalPatch *)blk, (EvalCam *)(blk + sizeof(EvalPatch)), win + (size_t)s * WS, st->ray, st->ref_cam, st->lod,
                         st->num_cam, st->cam_idx, lane);
    }
}
template <int NS, bool BYTES, bool ACCR>
__global__ PAIS_EVAL_BOUNDS(NS) void k_fitness(DevScene sc, const int32_t *stateIndex, const double *particles, double *out,
                                                                  int nEvals, int Kmax, const unsigned char *evalBlocks, size_t evalBlockBytes,
                                                                  const WinPix *win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    unsigned char *smem = smem0 + (threadIdx.x >> 6) * eval_lds_bytes(NS, Kmax, ACCR); // wave-private scratch
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *cbuf = Hbuf + Kmax * PAIS_H_STRIDE;
    const int lane = threadIdx.x & 63;
    const int WS = win_stride(sc);
    const int nw = (int)(eval_block_bytes(Kmax) / 8);
    for (int e = eval_first_task(); e < nEvals; e += eval_task_stride()) {
        const int s = stateIndex[e];
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)s);
        const uint64_t v0 = lane < nw ? src[lane] : 0, v1 = lane + 64 < nw ? src[lane + 64] : 0;
        wave_sync();
        stage_eval_block(smem, src, nw, lane, v0, v1);
        wave_sync();
        double f4[4], w4[4];
        const int bad = eval_fitness_parts<NS, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win + (size_t)s * WS, particles[3 * e], particles[3 * e + 1],
                                               particles[3 * e + 2], lane, 0, 1, f4, w4);
        if (lane == 0) out[e] = bad ? DBL_MAX : combine_parts(f4, w4);
    }
}

// ------------------------------------------- scalar setters on an LDS patch --
// All lanes of the (single-wave) workgroup execute these with identical data;
// stores are done by lane 0, callers __syncthreads() between phases.

__device__ bool cam_project(const DevScene &sc, int cam, const double *X, double *out, int LOD)
{
    const DevCamera &c = sc.cams[cam];
    project_raw(c.R, c.T, c.focal, c.pp, sc.lodScale[LOD], X, out);
    if (LOD > c.maxLOD) return false;
    return in_image_d(out, c.w[LOD], c.h[LOD]);
}

// Patch::setReferenceCameraIndex, patch.cpp:415-445
__device__ void set_reference_camera(const DevScene &sc, pais_patch_result *st, int lane)
{
    if (st->dropped) return;
    const int camNum = st->num_cam;
    int ref = -1, drop = 0;
    if (camNum < sc.cfg.minCamNum) {
        drop = 1;
        ref = st->ref_cam;
    } else {
        double maxCorr = -DBL_MAX;
        const double n0 = st->normal[0], n1 = st->normal[1], n2 = st->normal[2];
        for (int i = 0; i < camNum; i++) {
            const DevCamera &c = sc.cams[st->cam_idx[i]];
            double corr = n0 * (-c.optN[0]) + n1 * (-c.optN[1]) + n2 * (-c.optN[2]);
            if (corr > maxCorr) {
                maxCorr = corr;
                ref = st->cam_idx[i];
            }
        }
        if (ref < 0) {
            ref = st->cam_idx[0];
            drop = 1;
        }
    }
    __syncthreads();
    if (lane == 0) {
        st->ref_cam = ref;
        if (drop) st->dropped = 1;
    }
    __syncthreads();
}

// Patch::setDepthAndRay, patch.cpp:447-461
__device__ void set_depth_and_ray(const DevScene &sc, pais_patch_result *st, int lane)
{
    if (st->dropped) return;
    if (st->ref_cam < 0) {
        __syncthreads();
        if (lane == 0) st->dropped = 1;
        __syncthreads();
        return;
    }
    const DevCamera &rc = sc.cams[st->ref_cam];
    double ray[3];
    for (int i = 0; i < 3; ++i) ray[i] = st->center[i] - rc.C[i];
    double depth = norm3(ray);
    double inv = (1.0 / depth);
    __syncthreads();
    if (lane == 0) {
        st->depth = depth;
        for (int i = 0; i < 3; ++i) st->ray[i] = ray[i] * inv;
    }
    __syncthreads();
}

// Patch::setDepthRange, patch.cpp:463-509
__device__ void set_depth_range(const DevScene &sc, pais_patch_result *st, int lane)
{
    if (st->dropped) return;
    const int camNum = st->num_cam;
    if (camNum < sc.cfg.minCamNum) {
        __syncthreads();
        if (lane == 0) st->dropped = 1;
        __syncthreads();
        return;
    }
    const DevCamera &rc = sc.cams[st->ref_cam];
    double c1[3], c2[3];
    for (int i = 0; i < 3; ++i) {
        c1[i] = st->center[i];
        c2[i] = st->ray[i] * (st->depth + 1.0) + rc.C[i];
    }
    // per-camera image displacement in parallel, max() is order independent
    double wd = -DBL_MAX;
    for (int i = lane; i < camNum; i += 64) {
        const int ci = st->cam_idx[i];
        if (ci == st->ref_cam) continue;
        double p1[2], p2[2];
        cam_project(sc, ci, c1, p1, 0);
        cam_project(sc, ci, c2, p2, 0);
        double dx = p1[0] - p2[0], dy = p1[1] - p2[1];
        double imgDist = sqrt(dx * dx + dy * dy);
        double worldDist = 1.0 / imgDist;
        if (worldDist > wd && imgDist >= 0.01) wd = worldDist;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double o = __shfl_xor(wd, m, 64);
        wd = o > wd ? o : wd;
    }
    __syncthreads();
    if (lane == 0) {
        if (wd == -DBL_MAX) {
            st->dropped = 1;
        } else {
            double a = st->depth - wd * sc.cfg.depthRangeScalar;
            st->depthRange[0] = a > 0.0 ? a : 0.0;
            double b = wd * sc.cfg.depthRangeScalar, cc = sc.cfg.neighborRadius * 100;
            st->depthRange[1] = st->depth + (cc < b ? cc : b);
        }
    }
    __syncthreads();
}

// Patch::setLOD, patch.cpp:511-610
__device__ void set_lod(const DevScene &sc, pais_patch_result *st, int lane)
{
    if (st->dropped) return;
    if (st->ref_cam < 0) {
        __syncthreads();
        if (lane == 0) st->dropped = 1;
        __syncthreads();
        return;
    }
    const int r = sc.cfg.patchRadius, S = sc.cfg.patchSize, S2 = S * S;
    const DevCamera &rc = sc.cams[st->ref_cam];
    double c[3] = {st->center[0], st->center[1], st->center[2]};
    int LOD = sc.cfg.minLOD - 1;
    double variance = 0;
    while (variance < sc.cfg.textureVariation) {
        LOD++;
        if (LOD >= rc.maxLOD) {
            LOD = rc.maxLOD;
            break;
        }
        double pt[2];
        if (!cam_project(sc, st->ref_cam, c, pt, LOD)) {
            LOD = (LOD - 1) > 0 ? (LOD - 1) : 0;
            break;
        }
        const int cx = cv_round(pt[0]), cy = cv_round(pt[1]);
        const int w = rc.w[LOD], h = rc.h[LOD];
        if (cx - r < 0 || cx + r >= w || cy - r < 0 || cy + r >= h) { // inImage(x, y, LOD) for the whole window
            LOD = (LOD - 1) > 0 ? (LOD - 1) : 0;
            break;
        }
        const uint8_t *img = sc.imgBlob + rc.imgOff[LOD];
        int isum = 0;
        for (int k = lane; k < S2; k += 64) {
            int yi = k / S, xi = k - yi * S;
            isum += img[(cy - r + yi) * w + (cx - r + xi)];
        }
        isum = wave_sum_i(isum);
        double mean = (double)isum / (double)S2;
        double v = 0;
        for (int k = lane; k < S2; k += 64) {
            int yi = k / S, xi = k - yi * S;
            double t = (double)img[(cy - r + yi) * w + (cx - r + xi)];
            v += (t - mean) * (t - mean);
        }
        variance = wave_sum(v) / (double)S2;
    }
    __syncthreads();
    if (lane == 0) st->lod = LOD;
    __syncthreads();
}

// Patch::setPriority (patch.cpp:612-625) + Patch::setImagePoint (:627-653, colour is host-side)
__device__ void set_priority_and_image_point(const DevScene &sc, pais_patch_result *st, int lane)
{
    if (st->dropped) return;
    const int camNum = st->num_cam;
    double camRatio = ((double)camNum) / ((double)sc.numCams);
    double pri = st->fitness * det_exp(-st->correlation / 1.0 - camRatio / 1.0) * (st->lod + 1.0);
    double c[3] = {st->center[0], st->center[1], st->center[2]};
    __syncthreads();
    if (lane == 0) st->priority = pri;
    for (int i = lane; i < camNum; i += 64) {
        double p[2];
        cam_project(sc, st->cam_idx[i], c, p, 0);
        st->imgPoint[i][0] = p[0];
        st->imgPoint[i][1] = p[1];
    }
    __syncthreads();
}

// Patch::removeInvisibleCamera, patch.cpp:655-721 (with setCorrelationTable :221-267,
// getHomographyPatch :332-386; the region ratios of getHomographyRegionRatio :269-288 come from k_region_ratio).
//   hp     : global scratch of this workgroup, Kmax*S2 doubles (warped patches)
//   table  : LDS, Kmax*Kmax doubles ; Hn: LDS, Kmax*9 doubles ; flag: LDS, 1 int (drop)
//   ratios : global, this candidate's region ratio per visible camera
// Called by a workgroup of AFTER_WAVES waves: the cameras' warped patches are built by the waves in parallel (one camera
// per wave at a time), then the camera pairs of the correlation table are dealt to the waves; every value is produced
// by one wave with the operation sequence of the single-wave statement.  The rest runs redundantly in every wave with
// single-writer stores (thread 0).
#define AFTER_WAVES 4
__device__ void remove_invisible_camera(const DevScene &sc, pais_patch_result *st, double *hp, double *table,
                                        double *Hn, int *flag, double *ratios, int lane, int wave)
{
    if (st->dropped) return;
    const bool lead = (lane == 0) && (wave == 0);
    const int K = st->num_cam;
    const int r = sc.cfg.patchRadius, S = sc.cfg.patchSize, S2 = S * S;
    const int LOD = st->lod, refCam = st->ref_cam;
    const DevCamera &rc = sc.cams[refCam];
    const double s = sc.lodScale[LOD];
    double center[3] = {st->center[0], st->center[1], st->center[2]};
    double n[3] = {st->normal[0], st->normal[1], st->normal[2]};

    // getHomographies(center, normal, H)
    if (wave == 0) {
        const double d = -dot3(center, n);
        double Mref[9], invH[9];
        plane_matrix(d, s, rc.KR, rc.KT, n, Mref);
        inv3(Mref, invH);
        for (int c = lane; c < K; c += 64) {
            double H[9];
            const int ci = st->cam_idx[c];
            if (ci == refCam) {
                H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
            } else {
                double M[9];
                plane_matrix(d, s, sc.cams[ci].KR, sc.cams[ci].KT, n, M);
                mul33(M, invH, H);
            }
            for (int i = 0; i < 9; ++i) Hn[c * 9 + i] = H[i];
        }
    }
    for (int i = threadIdx.x; i < K * K; i += 64 * AFTER_WAVES) table[i] = 0;
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();

    double pt[2];
    cam_project(sc, refCam, center, pt, LOD);
    const double a0 = pt[0] - r, b0 = pt[1] - r;

    // setCorrelationTable: warped, L2-normalised patches, camera c by wave c mod AFTER_WAVES
    for (int c = wave; c < K; c += AFTER_WAVES) {
        const DevCamera &cam = sc.cams[st->cam_idx[c]];
        const uint8_t *img = sc.imgBlob + cam.imgOff[LOD];
        const int cw = cam.w[LOD], ch = cam.h[LOD];
        const double *H = Hn + 9 * c;
        double *hpc = hp + (size_t)c * S2;
        double sq = 0;
        bool bad = false;
        // no lane-dependent exit inside the walk (an out-of-image sample is flagged and sampled at a clamped position,
        // its value is never used: the whole patch is dropped), so the loads of the steps overlap
        for (int k = lane; k < S2; k += 64) {
            const int yi = k / S, xi = k - yi * S;
            const double x = a0 + (double)xi, y = b0 + (double)yi;
            const double w = (H[6] * x + H[7] * y + H[8]);
            const double ix = (H[0] * x + H[1] * y + H[2]) / w;
            const double iy = (H[3] * x + H[4] * y + H[5]) / w;
            const bool out = !(ix >= 0 && ix < cw - 1 && iy >= 0 && iy < ch - 1) || w == 0; // :355
            bad = bad || out;
            const double v = bilinear(img, cw, out ? 0.0 : ix, out ? 0.0 : iy);
            hpc[k] = v;
            sq += v * v;
        }
        if (__any(bad)) { // patch.cpp:243-247: any camera that leaves the image drops the patch
            if (lane == 0) atomicOr(flag, 1);
            continue;
        }
        sq = wave_sum(sq);
        const double inv = 1.0 / sqrt(sq);
        for (int k = lane; k < S2; k += 64) hpc[k] = hpc[k] * inv; // hp /= sqrt(sum)
    }
    __syncthreads(); // the warped patches of all waves are visible (same CU: global stores through the same L1 / L2)
    if (lead) st->ncc_tables += 1;
    if (*flag != 0) {
        // drop, correlation = 0 (and every later step is a no-op)
        __syncthreads();
        if (lead) {
            st->dropped = 1;
            st->correlation = 0;
        }
        __syncthreads();
        return;
    }
    {
        int p = 0;
        for (int i = 0; i < K; ++i) {
            for (int j = i + 1; j < K; ++j, ++p) {
                if (p % AFTER_WAVES != wave) continue; // uniform per wave
                const double *a = hp + (size_t)i * S2, *b = hp + (size_t)j * S2;
                double acc = 0;
                for (int k = lane; k < S2; k += 64) acc += a[k] * b[k];
                acc = wave_sum(acc);
                if (lane == 0) {
                    table[i * K + j] = acc;
                    table[j * K + i] = acc;
                }
            }
        }
    }
    __syncthreads();

    double correlation = 0;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) correlation += table[i * K + j];
    correlation /= (double)(K * K - K);

    double maxCorr = -DBL_MAX;
    int maxIdx = 0;
    for (int i = 0; i < K; ++i) {
        double corrSum = 0;
        for (int j = 0; j < K; ++j) corrSum += table[i * K + j];
        if (corrSum >= maxCorr) { // last max wins
            maxIdx = i;
            maxCorr = corrSum;
        }
    }

    // mark + erase, keeping order (removeIdx holds distinct camera indices): compacted in place by one thread
    __syncthreads();
    if (lead) {
        int nn = 0;
        for (int i = 0; i < K; ++i) {
            bool rem = false;
            const int ci = st->cam_idx[i];
            if (ratios[i] < sc.cfg.minRegionRatio) {
                rem = true;
            } else {
                const DevCamera &cam = sc.cams[ci];
                double dd = n[0] * (-cam.optN[0]) + n[1] * (-cam.optN[1]) + n[2] * (-cam.optN[2]);
                if (dd < 0) {
                    rem = true;
                } else if (i != maxIdx && table[maxIdx * K + i] < sc.cfg.minCorrelation) {
                    rem = true;
                }
            }
            if (!rem) {
                ratios[nn] = ratios[i]; // stays valid for the trailing removeInvisibleCamera if the homographies do
                st->cam_idx[nn++] = ci;
            }
        }
        st->correlation = correlation;
        st->num_cam = nn;
        if (nn < sc.cfg.minCamNum) st->dropped = 1;
    }
    __syncthreads();
}

// LDS <-> global copy of a patch record by one wave
__device__ void copy_record(pais_patch_result *dst, const pais_patch_result *src, int tid, int nthreads = 64)
{
    const uint32_t *s = (const uint32_t *)src;
    uint32_t *d = (uint32_t *)dst;
    for (int i = tid; i < (int)(sizeof(pais_patch_result) / 4); i += nthreads) d[i] = s[i];
}

// ---------------------------------------------------------------- k_begin ---
// candidate -> record; head of Patch::refine() (patch.cpp:117-136)
struct WinPix;
__device__ void pso_init_candidate(const DevScene &sc, const pais_patch_result *P, int c, unsigned char *states, int Nmax,
                                   int *activeList, int *activeCount, unsigned char *evalBlocks, size_t evalBlockBytes, WinPix *win,
                                   int lane);
// INIT: the wave goes straight on to the set-up of the candidate's first PSO run (k_pso_init's work) from the record it
// still holds in LDS -- one launch and one dependent round trip fewer at the head of every batch.  (The pass counters it
// adds to were cleared by the k_after<2> of the pass before: two sets used alternately, no memset launch per pass.)
template <bool INIT>
__global__ __launch_bounds__(64) void k_begin(DevScene sc, const pais_candidate *cands, pais_patch_result *recs, int n,
                                              unsigned char *states, int Nmax, int *activeList, int *activeCount,
                                              unsigned char *evalBlocks, size_t evalBlockBytes, WinPix *win)
{
    __shared__ pais_patch_result st;
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const pais_candidate *cd = &cands[c];
        __syncthreads();
        {
            uint32_t *d = (uint32_t *)&st;
            for (int i = lane; i < (int)(sizeof(pais_patch_result) / 4); i += 64) d[i] = 0;
        }
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < 3; ++i) {
                st.center[i] = cd->center[i];
                st.normal[i] = cd->normal[i];
            }
            st.normalS[0] = cd->normalS[0];
            st.normalS[1] = cd->normalS[1];
            st.key = cd->key;
            st.type = cd->type;
            int nc = cd->num_cam;
            st.num_cam = nc > PAIS_MAX_VIS ? PAIS_MAX_VIS : (nc < 0 ? 0 : nc);
            st.ref_cam = -1;
            st.lod = -1;
            st.fitness = DBL_MAX;   // abstractpatch.cpp:37-38
            st.priority = DBL_MAX;
            st.correlation = 0;
            st.stage = PAIS_STAGE_DONE;
        }
        __syncthreads();
        for (int i = lane; i < st.num_cam; i += 64) st.cam_idx[i] = cd->cam_idx[i];
        __syncthreads();

        if (st.num_cam < sc.cfg.minCamNum) { // patch.cpp:118-123
            if (lane == 0) {
                st.fitness = DBL_MAX;
                st.priority = DBL_MAX;
                st.dropped = 1;
            }
        } else {
            set_reference_camera(sc, &st, lane);
            set_depth_and_ray(sc, &st, lane);
            set_depth_range(sc, &st, lane);
            set_lod(sc, &st, lane);
            if (!st.dropped) {
                // patch.cpp:132-150: first evaluation of the while condition is always true
                __syncthreads();
                if (lane == 0) {
                    st.before_ref = st.ref_cam;
                    st.after_ref = -1;
                    st.before_num = st.num_cam;
                    st.after_num = -1;
                    st.total_cam_num = st.num_cam;
                    st.count = 1; // count++ evaluated once
                    st.stage = PAIS_STAGE_PSO;
                }
            }
        }
        __syncthreads();
        copy_record(&recs[c], &st, lane);
        if (INIT) pso_init_candidate(sc, &st, c, states, Nmax, activeList, activeCount, evalBlocks, evalBlockBytes, win, lane);
    }
}

// ------------------------------------- split PSO pipeline (large batches) ---
// The same GLN-PSO as k_pso, as a launch-per-iteration pipeline:
//   k_pso_init : per candidate, Patch::psoOptimization set-up + initParticles/setParticle
//   k_pso_eval : ONE WAVE PER (candidate, particle) -- PAIS::getFitness; no barriers, waves that
//                leave early (DBL_MAX particles) are simply replaced by the next pair
//   k_pso_step : per candidate, initFitness/updateFitness bookkeeping, updateGbest, inertia,
//                convergence test, moveParticles -- or the write-back when the run has ended
// Dependent launches cost ~2 us each against ~100 us of evaluation work per iteration, and the
// evaluation kernel runs at the throughput of the barrier-free k_fitness.
struct PsoState { // one per candidate, in global memory; the per-particle arrays follow (stride Nmax)
    double rangeL[3], rangeU[3], rangeInter[3], init[3];
    double iw, gBestFitness;
    uint64_t streamBase;
    int gIdx, N, maxIt, iteration, active, run, localK, started;
    // what the evaluation reads from the patch (patch.cpp:922-944)
    double ray[3];
    int refCam, LOD, K, pad;
    int camIdx[PAIS_MAX_VIS];
    // k_pso_iter: the loop-carried scalars of PsoSolver::run(), double buffered by launch parity
    struct IterDyn {
        double iw, gBestFitness;
        int gIdx, iteration, started, pad;
    } dyn[2];
};
#define PSO_DOUBLES_PER_PARTICLE (3 * 4 + 2 + 8)
__host__ __device__ inline size_t pso_state_bytes(int Nmax)
{
    // header + two swarm buffers (k_pso_iter reads one and writes the other; the other pipelines use buffer 0);
    // per particle 14 doubles of swarm state + 8 partial sums (evaluations shared by several waves)
    return ((sizeof(PsoState) + 15) & ~(size_t)15) + 2 * sizeof(double) * (size_t)Nmax * PSO_DOUBLES_PER_PARTICLE;
}
struct PsoArrays {
    double (*pos)[3], (*vec)[3], (*pBest)[3], (*nBest)[3];
    double *fit, *pBestFit;
    double (*part)[8]; // (f, w) of the 4 canonical sub-accumulators when an evaluation is shared by several waves
};
__device__ __forceinline__ PsoArrays pso_arrays(unsigned char *base, int Nmax, int buf = 0)
{
    PsoArrays a;
    unsigned char *q = base + ((sizeof(PsoState) + 15) & ~(size_t)15) + (size_t)buf * sizeof(double) * (size_t)Nmax * PSO_DOUBLES_PER_PARTICLE;
    a.pos = (double(*)[3])q; q += sizeof(double) * 3 * Nmax;
    a.vec = (double(*)[3])q; q += sizeof(double) * 3 * Nmax;
    a.pBest = (double(*)[3])q; q += sizeof(double) * 3 * Nmax;
    a.nBest = (double(*)[3])q; q += sizeof(double) * 3 * Nmax;
    a.fit = (double *)q; q += sizeof(double) * Nmax;
    a.pBestFit = (double *)q; q += sizeof(double) * Nmax;
    a.part = (double(*)[8])q;
    return a;
}

// activeList / activeCount (optional): compacted indices of the candidates that run a PSO in this pass, so that
// the per-iteration launches create waves only for them (k_pso_iter)
// evalBlocks / win: per candidate the evaluation block of the run (pais_eval.hpp): EvalPatch + EvalCam[M] exactly as the
// evaluation keeps them in LDS, and the reference window WinPix[S*S]; built ONCE per PSO run here, copied / read coalesced
// by every evaluation wave of the run
// set-up of ONE candidate's PSO run by one wave (P: its record -- in global memory, or the LDS copy k_begin just built)
__device__ void pso_init_candidate(const DevScene &sc, const pais_patch_result *P, int c, unsigned char *states, int Nmax,
                                   int *activeList, int *activeCount, unsigned char *evalBlocks, size_t evalBlockBytes, WinPix *win,
                                   int lane)
{
    const size_t SB = pso_state_bytes(Nmax);
    const int WS = win_stride(sc);
    PsoState *hd = (PsoState *)(states + SB * (size_t)c);
    PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
    if (P->stage != PAIS_STAGE_PSO || P->dropped) {
        if (lane == 0) { hd->active = 0; hd->started = 0; }
        return;
    }
    const int type = P->type;
    const int N = (type == PAIS_TYPE_SEED) ? sc.cfg.particleNum * 2 : sc.cfg.particleNum;
    const int maxIt = (type == PAIS_TYPE_SEED) ? sc.cfg.maxIteration * 2 : sc.cfg.maxIteration;
    // Patch::psoOptimization set-up (patch.cpp:183-200), identical in all lanes
    const double ns0 = P->normalS[0], ns1 = P->normalS[1];
    double L[3] = {0.0, ns1 - M_PI / 2.0, P->depthRange[0]};
    double U[3] = {M_PI, ns1 + M_PI / 2.0, P->depthRange[1]};
    if (type != PAIS_TYPE_SEED) {
        double lo = ns0 - M_PI / sc.cfg.reduceNormalRange, hi = ns0 + M_PI / sc.cfg.reduceNormalRange;
        L[0] = 0.0 < lo ? lo : 0.0;
        U[0] = hi < M_PI ? hi : M_PI;
        L[1] = ns1 - M_PI / sc.cfg.reduceNormalRange;
        U[1] = ns1 + M_PI / sc.cfg.reduceNormalRange;
    }
    const double init[3] = {ns0, ns1, P->depth};
    const uint64_t sb = stream_base(sc.seed, P->key);
    const uint32_t run = (uint32_t)P->pso_runs;
    if (lane == 0) {
        for (int d = 0; d < 3; ++d) {
            hd->rangeL[d] = L[d];
            hd->rangeU[d] = U[d];
            hd->rangeInter[d] = U[d] - L[d]; // psosolver.cpp:38
            hd->init[d] = init[d];
            hd->ray[d] = P->ray[d];
        }
        hd->N = N;
        hd->maxIt = maxIt;
        hd->localK = N < 5 ? N : 5; // psosolver.cpp:26
        hd->run = (int)run;
        hd->streamBase = sb;
        hd->iw = 0.8;
        hd->iteration = 0;
        hd->gIdx = 0;
        hd->gBestFitness = DBL_MAX;
        hd->active = 1;
        hd->started = 0;
        hd->dyn[0].iw = 0.8;
        hd->dyn[0].gBestFitness = DBL_MAX;
        hd->dyn[0].gIdx = 0;
        hd->dyn[0].iteration = 0;
        hd->dyn[0].started = 0;
        hd->refCam = P->ref_cam;
        hd->LOD = P->lod;
        hd->K = P->num_cam;
        if (activeList) activeList[atomicAdd(activeCount, 1)] = c;
    }
    for (int k = lane; k < P->num_cam; k += 64) hd->camIdx[k] = P->cam_idx[k];
    {
        unsigned char *blk = evalBlocks + evalBlockBytes * (size_t)c;
        build_eval_block(sc, (EvalPatch *)blk, (EvalCam *)(blk + sizeof(EvalPatch)), win + (size_t)c * WS, P->ray, P->ref_cam, P->lod,
                         P->num_cam, P->cam_idx, lane);
    }
    // initParticles (psosolver.cpp:94-110) + setParticle(init) (:267-284)
    for (int i = lane; i < N; i += 64) {
        for (int d = 0; d < 3; ++d) {
            const double ri = U[d] - L[d];
            const double u1 = uniform_from(sb, run, (uint32_t)(2 * (d * N + i)));
            const double u2 = uniform_from(sb, run, (uint32_t)(2 * (d * N + i) + 1));
            double p = (ri * u1) + L[d];
            double v = (2.0 * ri * u2) - ri;
            if (i == 0) {
                p = init[d];
                v = (2.0 * ri * uniform_from(sb, run, (uint32_t)(6 * N + d))) - ri;
            }
            A.pos[i][d] = p;
            A.vec[i][d] = v;
            A.pBest[i][d] = p;
            A.nBest[i][d] = 0; // particle.cpp:16
        }
    }
}
__global__ __launch_bounds__(64) void k_pso_init(DevScene sc, const pais_patch_result *recs, int n, unsigned char *states,
                                                 int Nmax, int *activeList, int *activeCount, unsigned char *evalBlocks,
                                                 size_t evalBlockBytes, WinPix *win)
{
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < n; c += gridDim.x)
        pso_init_candidate(sc, &recs[c], c, states, Nmax, activeList, activeCount, evalBlocks, evalBlockBytes, win, lane);
}

// Device-coherent access to the swarm state (k_pso_ring: the state of a candidate is handed from wave to wave INSIDE a launch,
// across XCDs with their own L2): agent-scope relaxed atomics compile to plain loads / stores with the sc1 bit -- served
// at the device's coherence point, no fence, no cache invalidation under the image taps.
// EXPERIMENT (-DPAIS_RING_XCD_SCOPE=1, profiles/r06_ring_xcd_scope.txt; NOT a product configuration): the ring's shared state at the
// scope of ONE XCD's L2 instead of the device -- loads as returning atomic ORs of 0 without sc1 (an atomic executes in the L2), stores
// as plain write-through stores, counters as workgroup-scope atomics.  Correct only while every wave that works a ring runs on the
// same XCD as the ring's other waves (workgroup b -> XCD b % 8 is how the dispatcher deals workgroups out today; nothing guarantees it).
#ifndef PAIS_RING_XCD_SCOPE
#define PAIS_RING_XCD_SCOPE 0
#endif
#if PAIS_RING_XCD_SCOPE
#define PAIS_RING_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ unsigned long long l2_load64(const void *p)
{
    unsigned long long r, z = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}
__device__ __forceinline__ unsigned l2_load32(const void *p)
{
    unsigned r, z = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}
__device__ __forceinline__ double cload(const double *p) { return __longlong_as_double((long long)l2_load64(p)); }
__device__ __forceinline__ void cstore(double *p, double v) { __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int cload(const int *p) { return (int)l2_load32(p); }
__device__ __forceinline__ void cstore(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned rload(const unsigned *p) { return l2_load32(p); }
__device__ __forceinline__ void rstore(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
#define PAIS_RING_SCOPE __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ double cload(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void cstore(double *p, double v)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int cload(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cstore(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rload(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rstore(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
template <bool COH, class T> __device__ __forceinline__ T sload(const T *p) { return COH ? cload(p) : *p; }
template <bool COH, class T> __device__ __forceinline__ void sstore(T *p, T v)
{
    if (COH) cstore(p, v);
    else *p = v;
}

#include "pais_pre.hpp"

// LDS of one swarm step: the swarm (Nmax * 14 doubles) + the scratch of swarm_eval_setup behind it
__host__ __device__ inline size_t step_lds_bytes(int Nmax) { return sizeof(double) * (size_t)Nmax * (3 * 4 + 2) + pre_scratch_bytes(Nmax); }

// The evaluation launch of large batches (split pipeline): one wave per (candidate, particle), nothing but the cost.
// The candidate's constants come from the block k_pso_init prepared; positions from swarm buffer 0 (k_pso_step).
template <int NS, bool BYTES, bool ACCR, bool PRE>
__global__ PAIS_EVAL_BOUNDS(NS) void k_pso_eval2(DevScene sc, unsigned char *states, int n, int Nmax, int Kmax,
                                                                    const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win,
                                                                    int pendingOnly, unsigned long long *verify, const double *pre)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    unsigned char *smem = smem0 + (threadIdx.x >> 6) * eval_lds_bytes(NS, Kmax, ACCR); // wave-private scratch
    const size_t preD = pre_rec_doubles(Kmax);
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *cbuf = Hbuf + Kmax * PAIS_H_STRIDE;
    const int lane = threadIdx.x & 63;
    const size_t SB = pso_state_bytes(Nmax);
    const int total = n * Nmax;
    const int WS = win_stride(sc);
    const int nwMax = (int)(eval_block_bytes(Kmax) / 8);
    for (int t = eval_first_task(); t < total; t += eval_task_stride()) {
        const int c = t / Nmax, i = t - c * Nmax; // candidate-major: a CU's waves gather from the same few image windows
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        // everything this wave needs from global memory is requested at once -- the run's state flags, the particle and
        // the first 128 words of the constants -- so that there is ONE memory round trip before the cost starts
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
        const int iLoad = i < Nmax ? i : 0;
        const int active = hd->active, Nrun = hd->N;
        const double p0 = A.pos[iLoad][0], p1 = A.pos[iLoad][1], p2 = A.pos[iLoad][2];
        const uint64_t v0 = lane < nwMax ? src[lane] : 0, v1 = lane + 64 < nwMax ? src[lane + 64] : 0;
        // pre: the particle's record {status, homographies} written by the swarm step (pais_pre.hpp) instead of its position
        const double *rec = PRE ? pre + preD * ((size_t)c * Nmax + iLoad) : nullptr;
        double status0 = 0.0, hv = 0.0;
        if (PRE) {
            status0 = rec[0];
            hv = lane < PAIS_H_STRIDE * Kmax ? rec[PAIS_PRE_HDR + lane] : 0.0;
        }
        // pendingOnly: the launch behind k_pso_tile (pais_tile.hpp) -- only the particles it flagged for the checked walk
        const double pend = pendingOnly ? A.part[iLoad][0] : 1.0;
        if (!active || i >= Nrun || (pend != 1.0 && pendingOnly != 2)) continue;
        wave_sync();
        stage_eval_block(smem, src, nwMax, lane, v0, v1);
        wave_sync();
        double f4[4], w4[4];
        int st;
        if (PRE) st = eval_fitness_pre<NS, BYTES, ACCR, false>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, rec, status0, hv, lane, f4, w4);
        else st = eval_fitness_parts<NS, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, p0, p1, p2, lane, 0, 1, f4, w4);
        if (lane == 0) {
            const double v = st ? DBL_MAX : combine_parts(f4, w4);
            // PAIS_TILE_VERIFY: the tile kernel's value against this walk's; mismatches are counted (and the first one kept)
            // in the debug words -- no printf in this kernel: its mere presence costs every instantiation registers
            if (pendingOnly == 2 && pend != 1.0 && A.fit[i] != v && verify) {
                if (atomicAdd(&verify[0], 1ULL) == 0) {
                    verify[1] = ((unsigned long long)(unsigned)c << 32) | (unsigned)i;
                    verify[2] = (unsigned long long)__double_as_longlong(A.fit[i]);
                    verify[3] = (unsigned long long)__double_as_longlong(v);
                }
            }
            A.fit[i] = v;
        }
    }
}

#include "pais_tile.hpp"
#include "pais_tile2.hpp"
#include "pais_literal.hpp"

// ------------------------------------------------------------- k_pso_iter ---
// Default PSO pipeline: ONE launch per PSO iteration.  The wave of (candidate c, particle i) first replays
// everything PsoSolver::run() does between two fitness passes (updateFitness, updateGbest, the convergence
// test, the inertia update: psosolver.cpp:121-149, 286-306) from the previous launch's swarm buffer -- the N
// particles of the candidate sit in lanes 0..N-1, so this costs a few hundred instructions instead of a
// dependent k_pso_step launch -- then moves ITS OWN particle (getLocalBest / setNearNeighborBest / moveParticles,
// :151-265), stores it into the other swarm buffer and evaluates the cost there.  Every wave of a candidate
// derives the same loop-carried scalars from the same inputs; particle 0's wave stores them (and, when the run
// ends, the result).  Comparisons and selections are reorganised over lanes, arithmetic is not: each value
// is produced by the same operation sequence as in pso_step_wave / oracle po_pso_run.

// lexicographic wave arg-min over (key, tie) of the lanes with valid != 0; returns the winner's tie or -1
__device__ __forceinline__ int wave_argmin_lex(bool valid, double key, int tie)
{
    double k = key;
    int t = tie;
    int v = valid ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ok = __shfl_xor(k, m, 64);
        const int ot = __shfl_xor(t, m, 64);
        const int ov = __shfl_xor(v, m, 64);
        const bool take = ov && (!v || ok < k || (ok == k && ot < t));
        k = take ? ok : k;
        t = take ? ot : t;
        v = take ? 1 : v;
    }
    return v ? t : -1;
}

// The same selection for a swarm that sits in lanes 0 .. N-1 of the first DPP row(s) (N <= 32: every swarm of an expansion
// candidate or seed with the README's particle count): the exchanges inside a row of 16 lanes are DPP operand modifiers of
// ordinary VALU moves (row_mirror, row_half_mirror, two quad permutations -- four symmetric pairings that together connect
// all 16 lanes) instead of ds_bpermute round trips through the LDS crossbar (~130 cycles each; the step replay of a latency-
// bound launch is a chain of four such selections: measured 16.6 k of the 34 k cycles of a thin launch's wave).  (key, tie)
// pairs are distinct, so any all-to-all pairing order selects the same winner.  Result of lane 0, broadcast.
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
// a permutation in which every lane has a source (rotations, mirrors): no `old` operand, so no copy in front of the move
template <int CTRL> __device__ __forceinline__ int dpp_get_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ void argmin_dpp_step(double &k, int &t, int &v)
{
    // (mirrors and quad permutations give every lane a source: the moves need no `old` operand, i.e. no copy in front of them)
    const int olo = dpp_get_i<CTRL>(__double2loint(k)), ohi = dpp_get_i<CTRL>(__double2hiint(k));
    const int ot = dpp_get_i<CTRL>(t), ov = dpp_get_i<CTRL>(v);
    const double ok = __hiloint2double(ohi, olo);
    const bool take = ov && (!v || ok < k || (ok == k && ot < t));
    k = take ? ok : k;
    t = take ? ot : t;
    v = take ? 1 : v;
}
__device__ __forceinline__ int wave_argmin_lex_n(bool valid, double key, int tie, int N)
{
    if (N > 32) return wave_argmin_lex(valid, key, tie);
    double k = key;
    int t = tie;
    int v = valid ? 1 : 0;
    if (N > 16) { // rows 0 and 1 first (one crossbar exchange)
        const double ok = __shfl_xor(k, 16, 64);
        const int ot = __shfl_xor(t, 16, 64), ov = __shfl_xor(v, 16, 64);
        const bool take = ov && (!v || ok < k || (ok == k && ot < t));
        k = take ? ok : k;
        t = take ? ot : t;
        v = take ? 1 : v;
    }
    argmin_dpp_step<0x140>(k, t, v); // row_mirror: lane i <-> 15 - i
    argmin_dpp_step<0x141>(k, t, v); // row_half_mirror: i <-> 7 - i
    argmin_dpp_step<0x1B>(k, t, v);  // quad_perm [3, 2, 1, 0]
    argmin_dpp_step<0xB1>(k, t, v);  // quad_perm [1, 0, 3, 2]
    return __builtin_amdgcn_readfirstlane(v ? t : -1);
}

// MEASUREMENT BUILDS ONLY (scripts/dup_profile.sh): PAIS_EXP_DUP = 1 ... 6 executes one component of k_pso_iter TWICE -- 1 the
// moveParticles selections / 2 the gBest scan + convergence sums / 3 the cost evaluation / 4 the four uniforms / 5 ranks + local
// best / 6 the three fitness-distance-ratio selections -- with identical results, so that the
// slowdown of the whole reconstruction is that component's share of the critical path (a cycle counter inside one wave is not:
// DESIGN 4.4).  The library is built with 0.
#ifndef PAIS_EXP_DUP
#define PAIS_EXP_DUP 0
#endif
__device__ __forceinline__ void exp_opaque(double &v) { asm volatile("" : "+v"(v)); }

// ---- lane-parallel forms of the step replay's serial scans (round 5; swarm of N <= 32 particles in lanes 0 .. N-1) ----------
// profiles/r05_dup_profile.txt: the replay's selections are 6.7 ms of the 80 ms pawn reconstruction's critical path (the
// moveParticles selections 3.9 ms, the gBest scan + convergence sums 2.8 ms), and the ISA shows why -- every `for (j < N)` over
// lane_get(v, j) is a loop of v_readlane_b32 with an SGPR lane index under exec-mask loop control (N came from a vector load),
// ~200 cycles per trip for one wave alone on its SIMD.  Below, each scan is a fixed number of DPP exchanges inside a 16-lane
// row (ordinary VALU moves with a lane-permuting operand modifier: no SGPR round trip, no loop, no branch), one crossbar
// exchange between rows 0 and 1 when N > 16.  Selections only -- every VALUE that is stored is produced by the same
// operation sequence as before, so the records are the same bits (the goldens and the parity suite did not move).
template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v)
{
    return __hiloint2double(dpp_mov_i<CTRL>(__double2hiint(v)), dpp_mov_i<CTRL>(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_get_d(double v)
{
    return __hiloint2double(dpp_get_i<CTRL>(__double2hiint(v)), dpp_get_i<CTRL>(__double2loint(v)));
}
// the butterfly of wave_argmin_lex_n as a plain sum: every lane of row 0 ends with the sum over lanes 0 .. 15 (N <= 16) or
// 0 .. 31 (rows 0 and 1).  The association order differs from lane to lane -- callers use lane 0's value (readfirstlane).
__device__ __forceinline__ double swarm_sum_fast(double v, int N)
{
    if (N > 16) v += __shfl_xor(v, 16, 64);
    v += dpp_get_d<0x140>(v); // row_mirror
    v += dpp_get_d<0x141>(v); // row_half_mirror
    v += dpp_get_d<0x1B>(v);  // quad_perm [3, 2, 1, 0]
    v += dpp_get_d<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    return lane_get(v, 0);
}
// `mean of 3 N terms < 0.01` (getDispersionIDX / getVelocityIDX, psosolver.cpp:70-92, 293-297) where the reference adds the
// terms one after the other.  t0, t1, t2: this lane's three terms (lanes >= N: anything).  A tree sum of the same terms differs
// from the sequential sum by < 3 N ulp (~1e-14 relative); when it is farther than 1e-9 from the threshold the comparison cannot
// depend on the order, otherwise (never seen) the sequential sum decides -- the decision is the reference's in every case.
__device__ __forceinline__ bool swarm_mean_below(double t0, double t1, double t2, int lane, int N)
{
    const double own = lane < N ? (t0 + t1) + t2 : 0.0;
    const double fast = swarm_sum_fast(own, N) / (double)(3 * N);
    if (fabs(fast - 0.01) > 1e-9) return fast < 0.01;
    double s = 0;
    for (int j = 0; j < N; ++j) {
        s += lane_get(t0, j);
        s += lane_get(t1, j);
        s += lane_get(t2, j);
    }
    s /= (double)(3 * N);
    return s < 0.01;
}
// updateGbest (psosolver.cpp:137-149): `for j: if (pBestFitness[j] <= gBestFitness) { gBestFitness = ...; gBest = j }` -- the
// scan ends at the LAST index of the swarm's minimum if that minimum is <= the incoming gBestFitness, and leaves (gf, g) alone
// otherwise (a NaN never satisfies `<=`, on either side).
__device__ __forceinline__ void swarm_update_gbest(double pbf, int lane, int N, double &gf, int &g)
{
    const int w = wave_argmin_lex_n(lane < N && pbf == pbf, pbf, 63 - lane, N);
    const int j = 63 - (w < 0 ? 63 : w);
    const double m = lane_get(pbf, j);
    if (w >= 0 && m <= gf) {
        gf = m;
        g = j;
    }
}
// rank(lane) = #{ j < N : d_j < d_lane or (d_j == d_lane and j < lane) } -- getLocalBest's stable sort position
// (psosolver.cpp:151-191) -- by rotating the distances through the 16-lane row (row_ror:n, n = 1 .. 15: every other lane of the
// row exactly once).  The source lane's index travels with its value, so the tie-break does not rest on the direction of the
// rotation.  Lanes N .. 31 must hold DBL_MAX.
template <int n> __device__ __forceinline__ int rank_ror(double dj, int lane)
{
    const double o = dpp_get_d<0x120 + n>(dj);
    const int j = dpp_get_i<0x120 + n>(lane);
    return (o < dj || (o == dj && j < lane)) ? 1 : 0;
}
template <int n> __device__ __forceinline__ int rank_ror_other(double dj, double od, bool hiRow)
{
    const double o = dpp_get_d<0x120 + n>(od);
    return (o < dj || (o == dj && hiRow)) ? 1 : 0;
}
__device__ __forceinline__ int swarm_rank(double dj, int lane, int N)
{
    int rank = 0;
    rank += rank_ror<1>(dj, lane);  rank += rank_ror<2>(dj, lane);  rank += rank_ror<3>(dj, lane);  rank += rank_ror<4>(dj, lane);
    rank += rank_ror<5>(dj, lane);  rank += rank_ror<6>(dj, lane);  rank += rank_ror<7>(dj, lane);  rank += rank_ror<8>(dj, lane);
    rank += rank_ror<9>(dj, lane);  rank += rank_ror<10>(dj, lane); rank += rank_ror<11>(dj, lane); rank += rank_ror<12>(dj, lane);
    rank += rank_ror<13>(dj, lane); rank += rank_ror<14>(dj, lane); rank += rank_ror<15>(dj, lane);
    if (N > 16) { // the other row's sixteen (rows 0 and 1 exchanged through the crossbar): every one of them precedes a lane of
                  // row 1, none a lane of row 0
        const double od = __shfl_xor(dj, 16, 64);
        const bool hiRow = (lane & 16) != 0;
        rank += (od < dj || (od == dj && hiRow)) ? 1 : 0;
        rank += rank_ror_other<1>(dj, od, hiRow);  rank += rank_ror_other<2>(dj, od, hiRow);  rank += rank_ror_other<3>(dj, od, hiRow);
        rank += rank_ror_other<4>(dj, od, hiRow);  rank += rank_ror_other<5>(dj, od, hiRow);  rank += rank_ror_other<6>(dj, od, hiRow);
        rank += rank_ror_other<7>(dj, od, hiRow);  rank += rank_ror_other<8>(dj, od, hiRow);  rank += rank_ror_other<9>(dj, od, hiRow);
        rank += rank_ror_other<10>(dj, od, hiRow); rank += rank_ror_other<11>(dj, od, hiRow); rank += rank_ror_other<12>(dj, od, hiRow);
        rank += rank_ror_other<13>(dj, od, hiRow); rank += rank_ror_other<14>(dj, od, hiRow); rank += rank_ror_other<15>(dj, od, hiRow);
    }
    return rank;
}

// FOUR lexicographic arg-mins at once (N <= 32): the selections of moveParticles -- the local best and the three per-dimension
// near-neighbour bests -- are independent chains of five dependent exchange-compare-select steps; written one after the other
// each waits for itself (1.5 us of the 13.3 us thin launch for the three FDR selections alone, profiles/r05_thin_launch_latency_
// budget.txt), interleaved step by step the four chains fill each other's latencies.  Same pairings, same comparisons as
// wave_argmin_lex_n: the same winners.
template <int CTRL> __device__ __forceinline__ void argmin4_dpp_step(double (&k)[4], int (&t)[4], int (&v)[4])
{
    int olo[4], ohi[4], ot[4], ov[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        olo[c] = dpp_get_i<CTRL>(__double2loint(k[c]));
        ohi[c] = dpp_get_i<CTRL>(__double2hiint(k[c]));
        ot[c] = dpp_get_i<CTRL>(t[c]);
        ov[c] = dpp_get_i<CTRL>(v[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double ok = __hiloint2double(ohi[c], olo[c]);
        const bool take = ov[c] && (!v[c] || ok < k[c] || (ok == k[c] && ot[c] < t[c]));
        k[c] = take ? ok : k[c];
        t[c] = take ? ot[c] : t[c];
        v[c] = take ? 1 : v[c];
    }
}
__device__ __forceinline__ void wave_argmin4_lex_n(const bool (&valid)[4], const double (&key)[4], const int (&tie)[4], int N, int (&out)[4])
{
    double k[4];
    int t[4], v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { k[c] = key[c]; t[c] = tie[c]; v[c] = valid[c] ? 1 : 0; }
    if (N > 16) { // rows 0 and 1 first (one crossbar exchange per word)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double ok = __shfl_xor(k[c], 16, 64);
            const int ot = __shfl_xor(t[c], 16, 64), ov = __shfl_xor(v[c], 16, 64);
            const bool take = ov && (!v[c] || ok < k[c] || (ok == k[c] && ot < t[c]));
            k[c] = take ? ok : k[c];
            t[c] = take ? ot : t[c];
            v[c] = take ? 1 : v[c];
        }
    }
    argmin4_dpp_step<0x140>(k, t, v); // row_mirror
    argmin4_dpp_step<0x141>(k, t, v); // row_half_mirror
    argmin4_dpp_step<0x1B>(k, t, v);  // quad_perm [3, 2, 1, 0]
    argmin4_dpp_step<0xB1>(k, t, v);  // quad_perm [1, 0, 3, 2]
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = __builtin_amdgcn_readfirstlane(v[c] ? t[c] : -1);
}

// moveParticles for particle i with the swarm in lanes (lane j = particle j, N <= 64): same selections as
// pso_move_particle (pais_dev.hpp), evaluated across lanes
__device__ __forceinline__ void pso_move_own(int i, int N, int localK, double iw, const double *u, const double *pos,
                                             const double *pb, double fitj, double pbf, int lane, const double *gB,
                                             const double *rl, const double *ru, const double *vecI, const double *nbI,
                                             double *outP, double *outV, double *outNb)
{
    const double pw = 1.2, gw = 1.5, lw = 1.0, nw = 1.0; // psosolver.h:110
    const double pVecW = pw * u[0], gVecW = gw * u[1], lVecW = lw * u[2], nVecW = nw * u[3];
    const bool pv = lane < N;
    const double pp0 = lane_get(pb[0], i), pp1 = lane_get(pb[1], i), pp2 = lane_get(pb[2], i);
    // getLocalBest: the localK nearest pBests by (squared distance, index); among them the first strict minimum
    // of pBestFitness in selection order
    double dj;
    {
        const double d0 = pp0 - pb[0], d1 = pp1 - pb[1], d2 = pp2 - pb[2];
        dj = 0;
        dj += d0 * d0;
        dj += d1 * d1;
        dj += d2 * d2;
        // (self: DBL_MAX, psosolver.cpp:161; lanes beyond the swarm: +inf, behind every particle in every order -- also behind
        //  one whose squared distance overflowed to +inf: equal keys are ordered by lane, and they have the higher lanes)
        dj = (lane == i) ? DBL_MAX : (!pv ? __builtin_inf() : dj);
    }
    int rank = 0;
    if (N <= 32) {
        rank = swarm_rank(dj, lane, N);
    } else {
        for (int j = 0; j < N; ++j) {
            const double o = lane_get(dj, j);
            rank += (o < dj || (o == dj && j < lane)) ? 1 : 0;
        }
    }
#if PAIS_EXP_DUP == 5
    {
        double dj2 = dj;
        exp_opaque(dj2);
        int rank2 = swarm_rank(dj2, lane, N);
        const int w2 = wave_argmin_lex_n(pv && rank2 < localK && pbf < DBL_MAX, pbf, rank2 * 64 + lane, N);
        asm volatile("" ::"s"(w2));
    }
#endif
    const bool sel = pv && rank < localK;
    // setNearNeighborBest: per dimension the first maximum of the fitness-distance ratio
    const double fitI = lane_get(fitj, i);
    if (N <= 32) {
        // the four selections side by side (wave_argmin4_lex_n)
        double FDR[3];
        for (int d = 0; d < 3; ++d) {
            const double pd = lane_get(pos[d], i);
            FDR[d] = (fitI - pbf) / fabs(pd - pb[d]);
        }
        const bool valid[4] = {sel && pbf < DBL_MAX, pv && lane != i && FDR[0] > -DBL_MAX, pv && lane != i && FDR[1] > -DBL_MAX,
                               pv && lane != i && FDR[2] > -DBL_MAX};
        const double key[4] = {pbf, -FDR[0], -FDR[1], -FDR[2]};
        const int tie[4] = {rank * 64 + lane, lane, lane, lane};
        int win[4];
        wave_argmin4_lex_n(valid, key, tie, N, win);
        const int lIdx4 = (win[0] < 0) ? i : (win[0] & 63);
        for (int d = 0; d < 3; ++d) {
            const int wn = win[1 + d];
            const double o = lane_get(pb[d], wn < 0 ? 0 : wn);
            outNb[d] = (wn < 0) ? nbI[d] : o;
        }
        for (int d = 0; d < 3; ++d) {
            double p = lane_get(pos[d], i);
            const double pbi = lane_get(pb[d], i), pbl = lane_get(pb[d], lIdx4);
            double v = iw * vecI[d] + pVecW * (pbi - p) + gVecW * (gB[d] - p) + lVecW * (pbl - p) + nVecW * (outNb[d] - p);
            outV[d] = v;
            p += v;
            if (p > ru[d]) p = ru[d];
            if (p < rl[d]) p = rl[d];
            outP[d] = p;
        }
        return;
    }
    const int w = wave_argmin_lex_n(sel && pbf < DBL_MAX, pbf, rank * 64 + lane, N);
    const int lIdx = (w < 0) ? i : (w & 63);
#if PAIS_EXP_DUP == 6
    for (int d = 0; d < 3; ++d) {
        double pd = lane_get(pos[d], i);
        exp_opaque(pd);
        const double FDR = (fitI - pbf) / fabs(pd - pb[d]);
        const int wn = wave_argmin_lex_n(pv && lane != i && FDR > -DBL_MAX, -FDR, lane, N);
        double o = lane_get(pb[d], wn < 0 ? 0 : wn);
        exp_opaque(o);
    }
#endif
    for (int d = 0; d < 3; ++d) {
        const double pd = lane_get(pos[d], i);
        const double FDR = (fitI - pbf) / fabs(pd - pb[d]);
        const bool cand = pv && lane != i && FDR > -DBL_MAX;
        const int wn = wave_argmin_lex_n(cand, -FDR, lane, N);
        const double o = lane_get(pb[d], wn < 0 ? 0 : wn);
        outNb[d] = (wn < 0) ? nbI[d] : o;
    }
    for (int d = 0; d < 3; ++d) {
        double p = lane_get(pos[d], i);
        const double pbi = lane_get(pb[d], i), pbl = lane_get(pb[d], lIdx);
        double v = iw * vecI[d] + pVecW * (pbi - p) + gVecW * (gB[d] - p) + lVecW * (pbl - p) + nVecW * (outNb[d] - p);
        outV[d] = v;
        p += v;
        if (p > ru[d]) p = ru[d];
        if (p < rl[d]) p = rl[d];
        outP[d] = p;
    }
}

// launch L = 0: cost of the initial swarm.  L >= 1: step (L-1) + cost of the moved particle.  finishOnly: one
// wave per candidate that only replays the step (the launch after the last possible iteration: every run ends).
// Tasks: positions [listLo, min(listHi, *activeCount)) of the active list written by k_pso_init.
// NS = 2 (3 waves per SIMD) for patches seen by few cameras, NS = 1 (4 waves per SIMD) beyond: a wave's LDS scratch
// grows with NS * K and caps the occupancy (measured: K ~ 4: NS 2 +3 %, K ~ 7: NS 1 +9 %)
template <int nparts, int NS, bool BYTES, bool ACCR>
__global__ PAIS_ITER_BOUNDS(nparts, NS) void k_pso_iter(DevScene sc, unsigned char *states, const int *activeList,
                                            const int *activeCount, int listLo, int listHi, int Nmax, int Kmax,
                                            pais_patch_result *recs, unsigned long long *stat, int L, int finishOnly,
                                            const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    unsigned char *smem = smem0 + (threadIdx.x >> 6) * eval_lds_bytes(NS, Kmax, ACCR); // wave-private scratch
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *cbuf = Hbuf + Kmax * PAIS_H_STRIDE;
    const int lane = threadIdx.x & 63;
    const size_t SB = pso_state_bytes(Nmax);
    const int WS = win_stride(sc);
    // nparts (1, 2, 4) consecutive waves share one evaluation: small rounds are bound by the latency of a
    // single evaluation wave, not by throughput.  Every part replays the step (identical results), part 0
    // stores the moved particle, each part stores its sub-accumulators; the consumer -- the step replay of
    // the next launch -- adds them in the canonical order.
    const int per = finishOnly ? 1 : Nmax * nparts;
    const int nAct = *activeCount;
    const int total = ((listHi < nAct ? listHi : nAct) - listLo) * per;
    for (int t = eval_first_task(); t < total; t += eval_task_stride()) {
        const int lp = t / per, ip = t - lp * per;
        const int c = activeList[listLo + lp];
        const int i = ip / nparts, part = ip - i * nparts;
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        // the candidate's evaluation constants are requested now and stored to LDS after the step replay: their memory
        // round trip overlaps the replay's
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
        const int nwMax = (int)(eval_block_bytes(Kmax) / 8);
        const uint64_t v0 = lane < nwMax ? src[lane] : 0, v1 = lane + 64 < nwMax ? src[lane + 64] : 0;
        // ... and so is everything the step replay reads: all addresses follow from c, i and the lane alone (particle
        // j of the swarm in lane j), so the run's flags, the swarm and the loop-carried scalars come back in ONE memory
        // round trip; a finished run wastes these loads
        PsoArrays Wb = pso_arrays((unsigned char *)hd, Nmax, L & 1);
        PsoArrays Rb = pso_arrays((unsigned char *)hd, Nmax, (L > 0 ? L - 1 : 0) & 1);
        const int jl = lane < Nmax ? lane : 0;
        // (wave-uniform by construction -- every lane loads the same word; said so, or the compiler keeps them in VGPRs and
        //  runs every loop over them under exec-mask control)
        const int active = __builtin_amdgcn_readfirstlane(hd->active), N = __builtin_amdgcn_readfirstlane(hd->N),
                  maxIt = __builtin_amdgcn_readfirstlane(hd->maxIt);
        PsoState::IterDyn dr = hd->dyn[(L > 0 ? L - 1 : 0) & 1];
        dr.started = __builtin_amdgcn_readfirstlane(dr.started);
        dr.iteration = __builtin_amdgcn_readfirstlane(dr.iteration);
        dr.gIdx = __builtin_amdgcn_readfirstlane(dr.gIdx);
        // (what only moveParticles reads is requested here too: behind the convergence test it would be a second round trip)
        const double rl[3] = {hd->rangeL[0], hd->rangeL[1], hd->rangeL[2]};
        const double ru[3] = {hd->rangeU[0], hd->rangeU[1], hd->rangeU[2]};
        const uint64_t streamBase = hd->streamBase;
        const int runIdx = __builtin_amdgcn_readfirstlane(hd->run), localK = __builtin_amdgcn_readfirstlane(hd->localK);
        double pos[3], pb[3];
        for (int d = 0; d < 3; ++d) {
            pos[d] = Rb.pos[jl][d];
            pb[d] = Rb.pBest[jl][d];
        }
        double fitj;
        if (nparts == 1) {
            fitj = Rb.fit[jl];
        } else {
            double f4[4], w4[4];
            bool bad = false;
            for (int a = 0; a < 4; ++a) {
                f4[a] = Rb.part[jl][2 * a];
                w4[a] = Rb.part[jl][2 * a + 1];
                bad = bad || (w4[a] < 0); // a part's call overflowed: DBL_MAX
            }
            fitj = bad ? DBL_MAX : combine_parts(f4, w4);
        }
        double pbf = Rb.pBestFit[jl];
        const double vecI[3] = {Rb.vec[i][0], Rb.vec[i][1], Rb.vec[i][2]};
        const double nbI[3] = {Rb.nBest[i][0], Rb.nBest[i][1], Rb.nBest[i][2]};
        const double q0 = Wb.pos[i][0], q1 = Wb.pos[i][1], q2 = Wb.pos[i][2]; // launch 0: the initial swarm
        if (!active || i >= N) continue;
        double p0, p1, p2;
        if (L == 0) {
            p0 = q0;
            p1 = q1;
            p2 = q2;
        } else {
            int it, g;
            double gf, iw;
            if (!dr.started) {
                // initFitness (:112-119) + run(): gBest = particles[0].pBest; updateGbest (:137-149)
                pbf = fitj;
                g = 0;
                gf = lane_get(pbf, 0);
                iw = dr.iw;
                it = 0;
            } else {
                // updateFitness (:121-135): pBest on strict '<'
                if (fitj < pbf) {
                    pbf = fitj;
                    pb[0] = pos[0];
                    pb[1] = pos[1];
                    pb[2] = pos[2];
                }
                g = dr.gIdx;
                gf = dr.gBestFitness;
                const double niw = dr.iw - 1.0 / maxIt; // :304
                iw = niw > 0.4 ? niw : 0.4;
                it = dr.iteration + 1;
            }
#if PAIS_EXP_DUP == 2
            { // the scan and the dispersion test once more (idempotent), inputs opaque
                double pbf2 = pbf, gf2 = gf;
                exp_opaque(pbf2);
                int g2 = g;
                swarm_update_gbest(pbf2, lane, N, gf2, g2);
                double q0 = pos[0];
                exp_opaque(q0);
                const double gB2[3] = {lane_get(pb[0], g2), lane_get(pb[1], g2), lane_get(pb[2], g2)};
                const bool b2 = swarm_mean_below(fabs(q0 - gB2[0]), fabs(pos[1] - gB2[1]), fabs(pos[2] - gB2[2]), lane, N);
                asm volatile("" ::"s"((int)b2));
                exp_opaque(gf2);
            }
#endif
#if PAIS_EXP_DUP == 12
            // (measurement build: the scan and the convergence test run twice AS A LOOP -- the same instructions a second time --
            //  where PAIS_EXP_DUP == 2 runs a second COPY of them: the difference is what fetching the copy's code costs)
            int expReps = 2;
            asm volatile("" : "+s"(expReps));
            const double gfIn = gf;
            const int gIn = g;
            double gB[3];
            bool finished = false;
#pragma nounroll
            for (int expRep = 0; expRep < expReps; ++expRep) {
            gf = gfIn;
            g = gIn;
            exp_opaque(pbf);
#endif
            if (N <= 32) {
                swarm_update_gbest(pbf, lane, N, gf, g);
            } else {
                for (int j = 0; j < N; ++j) {
                    const double v = lane_get(pbf, j);
                    if (v <= gf) {
                        gf = v;
                        g = j;
                    }
                }
            }
#if PAIS_EXP_DUP == 12
            gB[0] = lane_get(pb[0], g); gB[1] = lane_get(pb[1], g); gB[2] = lane_get(pb[2], g);
            finished = it >= maxIt;
#else
            const double gB[3] = {lane_get(pb[0], g), lane_get(pb[1], g), lane_get(pb[2], g)};
            // loop head of run(): `iteration < maxIteration`, then the convergence break (:293-297)
            bool finished = it >= maxIt;
#endif
            if (!finished) {
                const double a0 = fabs(pos[0] - gB[0]), a1 = fabs(pos[1] - gB[1]), a2 = fabs(pos[2] - gB[2]);
                bool dispBelow;
                if (N <= 32) {
                    dispBelow = swarm_mean_below(a0, a1, a2, lane, N);
                } else {
                    double disp = 0;
                    for (int j = 0; j < N; ++j) {
                        disp += lane_get(a0, j);
                        disp += lane_get(a1, j);
                        disp += lane_get(a2, j);
                    }
                    disp /= (double)(3 * N);
                    dispBelow = disp < 0.01;
                }
                if (dispBelow) {
                    const double v0 = fabs(Rb.vec[jl][0]), v1 = fabs(Rb.vec[jl][1]), v2 = fabs(Rb.vec[jl][2]);
                    if (N <= 32) {
                        finished = swarm_mean_below(v0, v1, v2, lane, N);
                    } else {
                        double vel = 0;
                        for (int j = 0; j < N; ++j) {
                            vel += lane_get(v0, j);
                            vel += lane_get(v1, j);
                            vel += lane_get(v2, j);
                        }
                        vel /= (double)(3 * N);
                        finished = vel < 0.01;
                    }
                }
            }
#if PAIS_EXP_DUP == 12
            }
#endif
            if (finished) {
                if (i == 0 && part == 0 && lane == 0) {
                    // write back (patch.cpp:208-213) and the maxFitness gate (:156-159)
                    pais_patch_result *P = &recs[c];
                    double nn[3];
                    spherical2normal(gB[0], gB[1], nn);
                    P->fitness = gf;
                    P->normalS[0] = gB[0];
                    P->normalS[1] = gB[1];
                    for (int q = 0; q < 3; ++q) P->normal[q] = nn[q];
                    P->depth = gB[2];
                    const DevCamera &rc = sc.cams[hd->refCam];
                    for (int q = 0; q < 3; ++q) P->center[q] = hd->ray[q] * gB[2] + rc.C[q];
                    P->pso_runs += 1;
                    P->pso_iterations += it;
                    const int evals = N * (1 + it);
                    P->pso_evals += evals;
                    if (gf > sc.cfg.maxFitness) {
                        P->dropped = 1;
                        P->stage = PAIS_STAGE_DONE;
                    } else {
                        P->stage = PAIS_STAGE_AFTER;
                    }
                    hd->active = 0;
                    const int K = hd->K;
                    const unsigned long long perEval = (unsigned long long)(4 * K + 1 + (sc.cfg.adaptiveDistanceEnable ? 8 : 0) +
                                                                            (sc.cfg.adaptiveGradientEnable ? 8 : 0));
                    atomicAdd(&stat[0], (unsigned long long)evals);
                    atomicAdd(&stat[1], (unsigned long long)evals * perEval);
                    atomicAdd(&stat[2], 1ULL);
                }
                continue;
            }
            // moveParticles (:220-265) for iteration `it`, own particle only
            double u[4];
            const uint32_t k0 = (uint32_t)(6 * N + 3 + 4 * (it * N + i));
            for (int q = 0; q < 4; ++q) u[q] = uniform_from(streamBase, (uint32_t)runIdx, k0 + q);
#if PAIS_EXP_DUP == 4
            {
                uint32_t k2 = k0;
                asm volatile("" : "+v"(k2));
                for (int q = 0; q < 4; ++q) { double t = uniform_from(streamBase, (uint32_t)runIdx, k2 + q); exp_opaque(t); }
            }
#endif
            double nP[3], nV[3], nNb[3];
#if PAIS_EXP_DUP == 1
            {
                double t = u[0];
                exp_opaque(t);
                const double u2[4] = {t, u[1], u[2], u[3]};
                pso_move_own(i, N, localK, iw, u2, pos, pb, fitj, pbf, lane, gB, rl, ru, vecI, nbI, nP, nV, nNb);
                exp_opaque(nP[0]); exp_opaque(nP[1]); exp_opaque(nP[2]); exp_opaque(nV[0]); exp_opaque(nNb[0]);
            }
#endif
            pso_move_own(i, N, localK, iw, u, pos, pb, fitj, pbf, lane, gB, rl, ru, vecI, nbI, nP, nV, nNb);
            const double pbI[3] = {lane_get(pb[0], i), lane_get(pb[1], i), lane_get(pb[2], i)};
            const double pbfI = lane_get(pbf, i);
            if (lane == 0 && part == 0) {
                for (int d = 0; d < 3; ++d) {
                    Wb.pos[i][d] = nP[d];
                    Wb.vec[i][d] = nV[d];
                    Wb.pBest[i][d] = pbI[d];
                    Wb.nBest[i][d] = nNb[d];
                }
                Wb.pBestFit[i] = pbfI;
                if (i == 0) {
                    PsoState::IterDyn dw;
                    dw.iw = iw;
                    dw.gBestFitness = gf;
                    dw.gIdx = g;
                    dw.iteration = it;
                    dw.started = 1;
                    dw.pad = 0;
                    hd->dyn[L & 1] = dw;
                }
            }
            p0 = nP[0];
            p1 = nP[1];
            p2 = nP[2];
        }
        if (finishOnly) continue; // unreachable for a well-formed schedule: every run has ended by now
        wave_sync();
        stage_eval_block(smem, src, nwMax, lane, v0, v1); // the run's evaluation block, prepared by k_pso_init
        wave_sync();
        double f4[4], w4[4];
#if PAIS_EXP_DUP == 3
        {
            double t = p0;
            exp_opaque(t);
            const int st2 = eval_fitness_parts<NS, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, t, p1, p2, lane, part, nparts, f4, w4);
            exp_opaque(f4[0]); exp_opaque(f4[1]); exp_opaque(f4[2]); exp_opaque(f4[3]);
            exp_opaque(w4[0]); exp_opaque(w4[1]); exp_opaque(w4[2]); exp_opaque(w4[3]);
            if (st2 < -5) continue;
            wave_sync();
        }
#endif
        const int st = eval_fitness_parts<NS, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, p0, p1, p2, lane, part, nparts, f4, w4);
        if (lane == 0) {
            if (nparts == 1) {
                Wb.fit[i] = st ? DBL_MAX : combine_parts(f4, w4);
            } else {
                for (int a = part; a < 4; a += nparts) {
                    Wb.part[i][2 * a] = st ? 0.0 : f4[a];
                    Wb.part[i][2 * a + 1] = st ? -1.0 : w4[a];
                }
            }
        }
    }
}

// Everything of PsoSolver::run() between two fitness passes, for ONE candidate, by one wave.
// smem: Nmax*(3*4+2) doubles of LDS scratch.
// Everything it reads was written by a previous launch.  Returns 1 if the run continues.
// COH: the state was written by another wave of THIS launch (k_pso_ring) and is read / written coherently
// pre != nullptr: the records of the moved swarm's evaluations (pais_pre.hpp) are written as well -- preEp / preCams: the
// candidate's evaluation block in LDS; pre: the candidate's first record; preD: doubles per record
template <bool COH = false>
__device__ int pso_step_wave(const DevScene &sc, pais_patch_result *recs, int c, PsoState *hd, int Nmax,
                               unsigned char *smem, unsigned long long *stat, int lane, const EvalPatch *preEp = nullptr,
                               const EvalCam *preCams = nullptr, double *pre = nullptr, size_t preD = 0)
{
    double(*pos)[3] = (double(*)[3])smem;
    double(*vec)[3] = pos + Nmax;
    double(*pBest)[3] = vec + Nmax;
    double(*nBest)[3] = pBest + Nmax;
    double *fit = (double *)(nBest + Nmax);
    double *pBestFit = fit + Nmax;
    PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
    const int N = hd->N, maxIt = hd->maxIt;
    __syncthreads();
    for (int i = lane; i < N; i += 64) {
        for (int d = 0; d < 3; ++d) {
            pos[i][d] = sload<COH>(&A.pos[i][d]);
            vec[i][d] = sload<COH>(&A.vec[i][d]);
            pBest[i][d] = sload<COH>(&A.pBest[i][d]);
            nBest[i][d] = sload<COH>(&A.nBest[i][d]);
        }
        fit[i] = sload<COH>(&A.fit[i]);
        pBestFit[i] = sload<COH>(&A.pBestFit[i]);
    }
    __syncthreads();
    int it = sload<COH>(&hd->iteration);
    int g = sload<COH>(&hd->gIdx);
    double gf = sload<COH>(&hd->gBestFitness);
    double iw = sload<COH>(&hd->iw);
    const int started = sload<COH>(&hd->started);
    if (!started) {
        // initFitness (:112-119) + run(): gBest = particles[0].pBest; updateGbest (:137-149)
        for (int i = lane; i < N; i += 64) pBestFit[i] = fit[i];
        __syncthreads();
        g = 0;
        gf = pBestFit[0];
        for (int j = 0; j < N; ++j)
            if (pBestFit[j] <= gf) { gf = pBestFit[j]; g = j; }
        it = 0;
    } else {
        // updateFitness (:121-135): pBest on strict '<'
        for (int i = lane; i < N; i += 64) {
            if (fit[i] < pBestFit[i]) {
                pBestFit[i] = fit[i];
                pBest[i][0] = pos[i][0];
                pBest[i][1] = pos[i][1];
                pBest[i][2] = pos[i][2];
            }
        }
        __syncthreads();
        for (int j = 0; j < N; ++j)
            if (pBestFit[j] <= gf) { gf = pBestFit[j]; g = j; }
        const double niw = iw - 1.0 / maxIt; // :304
        iw = niw > 0.4 ? niw : 0.4;
        it += 1;
    }
    // loop head of run(): `iteration < maxIteration`, then the convergence break (:293-297)
    bool finished = it >= maxIt;
    if (!finished) {
        const double g0 = pBest[g][0], g1 = pBest[g][1], g2 = pBest[g][2];
        double disp = 0;
        for (int i = 0; i < N; ++i) {
            disp += fabs(pos[i][0] - g0);
            disp += fabs(pos[i][1] - g1);
            disp += fabs(pos[i][2] - g2);
        }
        disp /= (double)(3 * N);
        if (disp < 0.01) {
            double vel = 0;
            for (int i = 0; i < N; ++i) {
                vel += fabs(vec[i][0]);
                vel += fabs(vec[i][1]);
                vel += fabs(vec[i][2]);
            }
            vel /= (double)(3 * N);
            finished = vel < 0.01;
        }
    }
    if (!finished) {
        // moveParticles (:220-265) for iteration `it`
        const double gB[3] = {pBest[g][0], pBest[g][1], pBest[g][2]};
        const double rl[3] = {hd->rangeL[0], hd->rangeL[1], hd->rangeL[2]};
        const double ru[3] = {hd->rangeU[0], hd->rangeU[1], hd->rangeU[2]};
        const uint64_t sb = hd->streamBase;
        const uint32_t run = (uint32_t)hd->run;
        const int localK = hd->localK;
        for (int i = lane; i < N; i += 64) {
            double u[4];
            const uint32_t k0 = (uint32_t)(6 * N + 3 + 4 * (it * N + i));
            for (int q = 0; q < 4; ++q) u[q] = uniform_from(sb, run, k0 + q);
            pso_move_particle(i, N, localK, iw, u, pos, vec, pBest, nBest, fit, pBestFit, gB, rl, ru);
        }
        __syncthreads();
        for (int i = lane; i < N; i += 64) {
            for (int d = 0; d < 3; ++d) {
                sstore<COH>(&A.pos[i][d], pos[i][d]);
                sstore<COH>(&A.vec[i][d], vec[i][d]);
                sstore<COH>(&A.pBest[i][d], pBest[i][d]);
                sstore<COH>(&A.nBest[i][d], nBest[i][d]);
            }
            sstore<COH>(&A.pBestFit[i], pBestFit[i]);
        }
        if (pre) swarm_eval_setup<COH>(sc, preEp, preCams, pos, N, pBestFit + Nmax, pre, preD, lane);
        if (lane == 0) {
            sstore<COH>(&hd->iteration, it);
            sstore<COH>(&hd->gIdx, g);
            sstore<COH>(&hd->gBestFitness, gf);
            sstore<COH>(&hd->iw, iw);
            sstore<COH>(&hd->started, 1);
        }
        return 1;
    } else if (lane == 0) {
        // write back (patch.cpp:208-213) and the maxFitness gate (:156-159)
        pais_patch_result *P = &recs[c];
        const double th = pBest[g][0], ph = pBest[g][1], dp = pBest[g][2];
        double nn[3];
        spherical2normal(th, ph, nn);
        P->fitness = gf;
        P->normalS[0] = th;
        P->normalS[1] = ph;
        for (int q = 0; q < 3; ++q) P->normal[q] = nn[q];
        P->depth = dp;
        const DevCamera &rc = sc.cams[hd->refCam];
        for (int q = 0; q < 3; ++q) P->center[q] = hd->ray[q] * dp + rc.C[q];
        P->pso_runs += 1;
        P->pso_iterations += it;
        const int evals = N * (1 + it);
        P->pso_evals += evals;
        if (gf > sc.cfg.maxFitness) {
            P->dropped = 1;
            P->stage = PAIS_STAGE_DONE;
        } else {
            P->stage = PAIS_STAGE_AFTER;
        }
        hd->active = 0;
        const int K = hd->K;
        const unsigned long long perEval = (unsigned long long)(4 * K + 1 + (sc.cfg.adaptiveDistanceEnable ? 8 : 0) +
                                                                (sc.cfg.adaptiveGradientEnable ? 8 : 0));
        atomicAdd(&stat[0], (unsigned long long)evals);
        atomicAdd(&stat[1], (unsigned long long)evals * perEval);
        atomicAdd(&stat[2], 1ULL);
        atomicAdd(&stat[5], (unsigned long long)evals); // ... of which by the large-batch pipeline (k_pso_eval2 / k_pso_ring)
        atomicAdd(&stat[6], (unsigned long long)evals * perEval);
        if (COH) { // ... of which inside a k_pso_ring launch
            atomicAdd(&stat[22], (unsigned long long)evals);
            atomicAdd(&stat[23], (unsigned long long)evals * perEval);
        }
    }
    return 0;
}
// the step as k_pso_ring calls it.  PAIS_RING_STEP_CALL 1: a real function call, so that the step's registers are not part of
// the evaluation loop's allocation problem.  Measured and NOT kept (profiles/r04_ring_step_call_ab.txt):
// pawn 85.8-88.1 ms against 82.1-82.5 ms inlined -- the call's frame (368-416 B of scratch per lane: the ABI's callee-saved
// registers) costs more than the 96 B of the inlined build, whose scratch accesses all sit outside the tap loops (entry: 14
// stores of kernel arguments; ~7 reloads per evaluation next to ~3 000 VALU instructions; the rest inside the step itself)
#ifndef PAIS_RING_STEP_CALL
#define PAIS_RING_STEP_CALL 0
#endif
#if PAIS_RING_STEP_CALL
__attribute__((noinline))
#endif
__device__ int pso_step_wave_ring(const DevScene &sc, pais_patch_result *recs, int c, PsoState *hd, int Nmax, unsigned char *smem,
                                  unsigned long long *stat, int lane, const EvalPatch *preEp, const EvalCam *preCams, double *pre, size_t preD)
{
    return pso_step_wave<true>(sc, recs, c, hd, Nmax, smem, stat, lane, preEp, preCams, pre, preD);
}
// the step kernel of large batches: one wave per candidate
// pre != nullptr: the step also writes the evaluation records of the moved swarm (pais_pre.hpp) from the candidate's evaluation
// block, which it stages in front of its scratch
__global__ __launch_bounds__(64) void k_pso_step(DevScene sc, pais_patch_result *recs, unsigned char *states, int n,
                                                 int Nmax, unsigned long long *stat, const unsigned char *evalBlocks, size_t evalBlockBytes,
                                                 double *pre, int Kmax)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    unsigned char *smem = smem0 + (pre ? evalBlockBytes : 0);
    const size_t preD = pre_rec_doubles(Kmax);
    const int nwMax = (int)(evalBlockBytes / 8);
    // the step of one slice runs while the other slice's evaluation launch fills the GPU; its few waves sit on the
    // critical path of their own slice (next evaluation launch), so they take issue priority over the evaluation waves
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const size_t SB = pso_state_bytes(Nmax);
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        if (!hd->active) continue;
        if (pre) {
            const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
            wave_sync();
            stage_eval_block(smem0, src, nwMax, lane, lane < nwMax ? src[lane] : 0, lane + 64 < nwMax ? src[lane + 64] : 0);
            wave_sync();
        }
        pso_step_wave(sc, recs, c, hd, Nmax, smem, stat, lane, (const EvalPatch *)smem0, (const EvalCam *)(smem0 + sizeof(EvalPatch)),
                      pre ? pre + preD * (size_t)Nmax * (size_t)c : nullptr, preD);
    }
}

// the records of the INITIAL swarm (k_begin / k_pso_init wrote the positions and the evaluation block): one wave per candidate
__global__ __launch_bounds__(64) void k_pso_setup0(DevScene sc, unsigned char *states, int n, int Nmax, const unsigned char *evalBlocks,
                                                   size_t evalBlockBytes, double *pre, int Kmax)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    double(*pos)[3] = (double(*)[3])(smem0 + evalBlockBytes);
    double *scr = (double *)(pos + Nmax);
    const int lane = threadIdx.x;
    const size_t SB = pso_state_bytes(Nmax);
    const size_t preD = pre_rec_doubles(Kmax);
    const int nwMax = (int)(evalBlockBytes / 8);
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        if (!hd->active) continue;
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        const int N = hd->N;
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
        wave_sync();
        stage_eval_block(smem0, src, nwMax, lane, lane < nwMax ? src[lane] : 0, lane + 64 < nwMax ? src[lane + 64] : 0);
        for (int i = lane; i < N; i += 64)
            for (int d = 0; d < 3; ++d) pos[i][d] = A.pos[i][d];
        wave_sync();
        swarm_eval_setup<false>(sc, (const EvalPatch *)smem0, (const EvalCam *)(smem0 + sizeof(EvalPatch)), pos, N, scr,
                                pre + preD * (size_t)Nmax * (size_t)c, preD, lane);
    }
}

// ---------------------------------------------------------------- k_pso_ring ---
// A PSO pass of a large batch WITHOUT per-iteration launches.  Iteration i + 1 of a candidate depends on that candidate's N
// evaluations of iteration i and on nothing else, so the pass is a pool of tasks -- (candidate, particle): evaluate the
// particle where it stands -- on a ring in global memory, worked off by a fixed set of resident waves:
//   * a wave takes the next ring index (one atomic), waits until that entry is published, evaluates, stores the fitness
//     and counts itself in at the candidate (one atomic);
//   * the wave that completes a candidate's count runs the candidate's swarm step (pso_step_wave: the code of k_pso_step)
//     and publishes the N tasks of the next iteration at the ring's tail -- or the result, if the run has ended.
// Nobody waits for a particular wave: an index beyond the published tail is only ever held by a wave that would otherwise be
// idle, and the tasks whose completion will publish it are being executed by waves that are not waiting -- no deadlock, with
// any number of resident waves.  The swarm state crosses waves (and XCDs) through sc1 loads / stores (cload / cstore above);
// a payload is complete (s_waitcnt vmcnt(0)) before the atomic that lets another wave look at it.  Every wait is bounded in
// WALL TIME (s_memrealtime, the 100 MHz constant counter: a poll count would depend on the clock, a profiler's serialisation
// or a neighbour on a shared GPU): a wave that has waited `timeoutTicks` raises the error word and leaves; the host then
// re-runs the batch through the per-iteration launches (pais_capi.hip ring_fallback) -- the records are the same.
// Same evaluation code, same step code, per-candidate order of operations unchanged: the records are those of
// k_pso_eval2 + k_pso_step bit for bit.
#define PAIS_RING_EMPTY 0xFFFFFFFFu
static_assert(PAIS_WG_WAVES == 1, "k_pso_ring: the swarm step (pso_step_wave) synchronises with workgroup barriers inside wave-divergent "
                                  "control flow, which is only a wave barrier while a workgroup is ONE wave");
// PAIS_RINGS (pais_internal.h): a multiple of 8 -- workgroup b runs on XCD b % 8 and works ring b % PAIS_RINGS, so a ring's
// counters and its candidates' state stay in one XCD's L2
// two 64-byte lines per ring: the head counter (one atomic per task, from every wave) alone on the first; what publishers and
// idle waves touch (tail, done, error) on the second -- polling waves do not pull the line the poppers serialise on
struct RingCtl { unsigned head, pad0[15]; unsigned tail, done, total, error, pad1[12]; };
static_assert(sizeof(RingCtl) == PAIS_RING_CTL_BYTES, "RingCtl layout (pais_internal.h)");

// candidates c with c % PAIS_RINGS == r belong to ring r; its segment of the ring memory starts at r * segWords
// tasks of iteration 0 (the initial swarm); candidates whose refinement ended in k_begin count as done
__global__ __launch_bounds__(256) void k_ring_init(unsigned char *states, int n, int Nmax, unsigned *ring, unsigned segWords, RingCtl *ctl)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    RingCtl *rc = &ctl[c % PAIS_RINGS];
    atomicAdd(&rc->total, 1u);
    const PsoState *hd = (const PsoState *)(states + pso_state_bytes(Nmax) * (size_t)c);
    if (!hd->active) {
        atomicAdd(&rc->done, 1u);
        return;
    }
    const int N = hd->N;
    const unsigned base = atomicAdd(&rc->tail, (unsigned)N);
    unsigned *seg = ring + (size_t)(c % PAIS_RINGS) * segWords;
    for (int j = 0; j < N; ++j) seg[base + j] = ((unsigned)c << 8) | (unsigned)j;
}

// MEASUREMENT BUILDS ONLY (-DPAIS_RING_PROFILE=1, profiles/r06_ring_group_ab.txt): where a ring wave's time goes -- s_memtime at the phase
// boundaries of every task, summed over all waves: [0] index (head atomic) [1] entry wait [2] loads + staging [3] evaluation
// [4] fitness store + arrival [5] swarm step + publish [6] tasks [7] steps; printed when a context is destroyed
#ifndef PAIS_RING_PROFILE
#define PAIS_RING_PROFILE 0
#endif
#if PAIS_RING_PROFILE
__device__ unsigned long long g_ringProf[8];
#define PAIS_RP_MARK(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#else
#define PAIS_RP_MARK(v)
#endif
template <int NS, bool BYTES, bool ACCR, bool PRE>
__global__ PAIS_EVAL_BOUNDS(NS) void k_pso_ring(DevScene sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax, int Kmax,
                                                const unsigned char *evalBlocks, size_t evalBlockBytes, const WinPix *win, unsigned *ringAll,
                                                unsigned segWords, RingCtl *ctlAll, int *arrive, unsigned long long *stat, size_t ldsPerWave,
                                                unsigned long long timeoutTicks, double *pre)
{
    const size_t preD = pre_rec_doubles(Kmax);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
    unsigned char *smem = smem0 + (threadIdx.x >> 6) * ldsPerWave; // wave-private scratch (evaluation; the step reuses it)
    EvalPatch *ep = (EvalPatch *)smem;
    EvalCam *cams = (EvalCam *)(smem + sizeof(EvalPatch));
    double *Hbuf = (double *)(smem + eval_block_bytes(Kmax));
    double *cbuf = Hbuf + Kmax * PAIS_H_STRIDE;
    const int lane = threadIdx.x & 63;
    const size_t SB = pso_state_bytes(Nmax);
    const int WS = win_stride(sc);
    const int nwMax = (int)(eval_block_bytes(Kmax) / 8);
    const int myRing = (int)(blockIdx.x % PAIS_RINGS);
    RingCtl *ctl = &ctlAll[myRing];
    unsigned *ring = ringAll + (size_t)myRing * segWords;
    const unsigned cap = segWords;
    const unsigned total = ctl->total; // (written by k_ring_init, before this launch)
    for (;;) {
        PAIS_RP_MARK(rpA)
        unsigned idx = 0;
        if (lane == 0) idx = __hip_atomic_fetch_add(&ctl->head, 1u, __ATOMIC_RELAXED, PAIS_RING_SCOPE);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx >= cap) break;
        PAIS_RP_MARK(rpB)
        unsigned e = PAIS_RING_EMPTY;
        unsigned long long t0 = 0;
        for (int spins = 0;; ++spins) {
            e = rload(&ring[idx]);
            if (e != PAIS_RING_EMPTY) break;
            if ((spins & 7) == 7) { // (the shared words are looked at now and then: thousands of idle waves poll)
                if (rload(&ctl->done) >= total) break; // every run of this ring has ended
                if (rload(&ctl->error) != 0) break;
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (spins == 7) t0 = now;
                if (now - t0 >= timeoutTicks) { // nobody has published this entry for `timeoutTicks` of wall time
                    if (lane == 0) rstore(&ctl->error, 1u);
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(32);
        }
        e = (unsigned)__builtin_amdgcn_readfirstlane((int)e);
        if (e == PAIS_RING_EMPTY) break;
        PAIS_RP_MARK(rpC)
        const int c = (int)(e >> 8), i = (int)(e & 255u);
        PsoState *hd = (PsoState *)(states + SB * (size_t)c);
        PsoArrays A = pso_arrays((unsigned char *)hd, Nmax);
        const uint64_t *src = (const uint64_t *)(evalBlocks + evalBlockBytes * (size_t)c);
        // pre: the particle's record {status, homographies} written by the swarm step / k_pso_setup0 (pais_pre.hpp) instead of its position
        double *crec = PRE ? pre + preD * (size_t)Nmax * (size_t)c : nullptr;
        const double *rec = PRE ? crec + preD * (size_t)i : nullptr;
        double p0 = 0, p1 = 0, p2 = 0, status0 = 0, hv = 0;
        if (PRE) {
            status0 = cload(&rec[0]);
            hv = lane < PAIS_H_STRIDE * Kmax ? cload(&rec[PAIS_PRE_HDR + lane]) : 0.0;
        } else {
            p0 = cload(&A.pos[i][0]); p1 = cload(&A.pos[i][1]); p2 = cload(&A.pos[i][2]);
        }
        const uint64_t v0 = lane < nwMax ? src[lane] : 0, v1 = lane + 64 < nwMax ? src[lane + 64] : 0;
        const int N = hd->N; // (written by k_begin, before this launch)
        wave_sync();
        stage_eval_block(smem, src, nwMax, lane, v0, v1);
        wave_sync();
#if PAIS_RING_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PAIS_RP_MARK(rpD)
        double f4[4], w4[4];
        int st;
        if (PRE) st = eval_fitness_pre<NS, BYTES, ACCR, true>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, rec, status0, hv, lane, f4, w4);
        else st = eval_fitness_parts<NS, BYTES, ACCR>(sc, ep, cams, Hbuf, cbuf, win + (size_t)c * WS, p0, p1, p2, lane, 0, 1, f4, w4);
        int old = 0;
        PAIS_RP_MARK(rpE)
        if (lane == 0) {
            cstore(&A.fit[i], st ? DBL_MAX : combine_parts(f4, w4));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the fitness is at the coherence point before it is counted
            old = __hip_atomic_fetch_add(&arrive[(size_t)c * PAIS_ARRIVE_STRIDE], 1, __ATOMIC_RELAXED, PAIS_RING_SCOPE);
        }
        old = __builtin_amdgcn_readfirstlane(old);
#if PAIS_RING_PROFILE
        {
            const unsigned long long rpF = __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                atomicAdd(&g_ringProf[0], rpB - rpA); atomicAdd(&g_ringProf[1], rpC - rpB); atomicAdd(&g_ringProf[2], rpD - rpC);
                atomicAdd(&g_ringProf[3], rpE - rpD); atomicAdd(&g_ringProf[4], rpF - rpE); atomicAdd(&g_ringProf[6], 1ULL);
            }
        }
        PAIS_RP_MARK(rpG)
#endif
        if (old != N - 1) continue;
        // this wave completed the candidate's iteration: its swarm step, then the next iteration's tasks (or the result)
        __builtin_amdgcn_s_setprio(3); // the step sits on the candidate's critical path
        if (lane == 0) cstore(&arrive[(size_t)c * PAIS_ARRIVE_STRIDE], 0);
        wave_sync();
        // (with records: the step's scratch lies behind the evaluation block, which swarm_eval_setup reads -- this wave has just
        //  evaluated a particle of candidate c, so the block in its LDS is c's)
        const int cont = pso_step_wave_ring(sc, recs, c, hd, Nmax, smem + (PRE ? eval_block_bytes(Kmax) : 0), stat, lane, ep, cams, crec, preD);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every lane's part of the new swarm (and the counter reset) is out
        wave_sync();
        if (__builtin_amdgcn_readfirstlane(cont)) {
            unsigned base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add(&ctl->tail, (unsigned)N, __ATOMIC_RELAXED, PAIS_RING_SCOPE);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (base + (unsigned)N > cap) { // cannot happen (a segment holds every task of its candidates); never write past it
                if (lane == 0) rstore(&ctl->error, 2u);
                break;
            }
            if (lane < N) rstore(&ring[base + lane], ((unsigned)c << 8) | (unsigned)lane);
        } else if (lane == 0) {
            __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, PAIS_RING_SCOPE);
        }
        __builtin_amdgcn_s_setprio(0);
#if PAIS_RING_PROFILE
        {
            const unsigned long long rpH = __builtin_amdgcn_s_memtime();
            if (lane == 0) { atomicAdd(&g_ringProf[5], rpH - rpG); atomicAdd(&g_ringProf[7], 1ULL); }
        }
#endif
    }
}

// ---------------------------------------------------------------- k_after ---
// After one PSO run: patch.cpp:161-175 (+ the caller's removeInvisibleCamera, mvs.cpp:215 / :574, once the refine loop
// has ended).  Four launches:
//   k_region_ratio(AFTER)   region ratio of every (candidate, visible camera): ONE LANE each -- three small Jacobi SVDs
//                           (cv::fitEllipse) are a long serial chain, so they are packed 64 to a wave across candidates
//                           instead of occupying K lanes of a candidate's wave
//   k_after<1>              first removeInvisibleCamera, setters, seed-loop control, priority + image points
//   k_region_ratio(AFTER2)  the same for the state after the setters (other reference camera / LOD / camera set)
//   k_after<2>              the trailing removeInvisibleCamera
__global__ __launch_bounds__(64) void k_region_ratio(DevScene sc, const pais_patch_result *recs, int n, int stage, double *ratios, int Kmax)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int c = t / Kmax, k = t - c * Kmax; // Kmax = the batch's largest camera count: (nearly) every lane has a camera
    if (c >= n) return;
    const pais_patch_result *P = &recs[c];
    if (P->stage != stage || P->dropped || k >= P->num_cam) return;
    const int LOD = P->lod, refCam = P->ref_cam;
    const DevCamera &rc = sc.cams[refCam];
    const double s = sc.lodScale[LOD];
    const double center[3] = {P->center[0], P->center[1], P->center[2]};
    const double nrm[3] = {P->normal[0], P->normal[1], P->normal[2]};
    double H[9];
    const int ci = P->cam_idx[k];
    if (ci == refCam) {
        H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1;
    } else {
        const double d = -dot3(center, nrm);
        double Mref[9], invH[9], M[9];
        plane_matrix(d, s, rc.KR, rc.KT, nrm, Mref);
        inv3(Mref, invH);
        plane_matrix(d, s, sc.cams[ci].KR, sc.cams[ci].KT, nrm, M);
        mul33(M, invH, H);
    }
    double pt[2];
    cam_project(sc, refCam, center, pt, LOD);
    ratios[(size_t)c * PAIS_MAX_VIS + k] = region_ratio(pt[0], pt[1], sc.cfg.patchRadius, H);
}

template <int PHASE>
__global__ __launch_bounds__(64 * AFTER_WAVES) void k_after(DevScene sc, pais_patch_result *recs, int n, double *hpScratch,
                                                              int *counters, unsigned long long *stat, int Kmax, double *ratios,
                                                              int hpInLds, int *nextCounters)
{
    // the counters the NEXT pass (of this or the next batch) will add to; nothing of this pass touches that set
    if (PHASE == 2 && nextCounters && blockIdx.x == 0 && threadIdx.x == 0) {
        nextCounters[0] = 0;
        nextCounters[1] = 0;
        nextCounters[2] = 0;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pais_patch_result *st = (pais_patch_result *)smem;
    size_t off = (sizeof(pais_patch_result) + 15) & ~(size_t)15;
    double *table = (double *)(smem + off); off += sizeof(double) * Kmax * Kmax;
    double *Hn = (double *)(smem + off); off += sizeof(double) * 9 * Kmax;
    int *flag = (int *)(smem + off); off += 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool lead = threadIdx.x == 0;
    const int S2 = sc.cfg.patchSize * sc.cfg.patchSize;
    // the warped patches (Kmax x S2 doubles, written once, read K times): LDS while they fit, else a global scratch slab
    double *hp = hpInLds ? (double *)(smem + off) : hpScratch + (size_t)blockIdx.x * Kmax * S2;
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const int stg = recs[c].stage;
        if (PHASE == 1 ? (stg != PAIS_STAGE_AFTER) : (stg != PAIS_STAGE_AFTER2 && stg != PAIS_STAGE_AFTER2_KEEP && stg != PAIS_STAGE_AFTER2_SAME))
            continue;
        __syncthreads();
        copy_record(st, &recs[c], threadIdx.x, 64 * AFTER_WAVES);
        __syncthreads();

        const int nccBefore = st->ncc_tables;
        double *rat = ratios + (size_t)c * PAIS_MAX_VIS;
        if (PHASE == 1) {
            const int refBefore = st->ref_cam, lodBefore = st->lod, numBefore = st->num_cam;
            remove_invisible_camera(sc, st, hp, table, Hn, flag, rat, lane, wave);
            const bool removedNone = st->num_cam == numBefore; // (read behind the call's closing barrier, by every thread alike)
            set_reference_camera(sc, st, lane);
            set_depth_and_ray(sc, st, lane);
            set_depth_range(sc, st, lane);
            set_lod(sc, st, lane);

            bool again = false;
            if (st->type != PAIS_TYPE_EXPAND && !st->dropped) {
                // patch.cpp:170-171 then the while condition of :140
                const int afterRef = st->ref_cam, afterNum = st->num_cam;
                const int cnt = st->count;
                const bool cond = (st->before_ref != afterRef || st->before_num != afterNum) && (cnt <= st->total_cam_num);
                __syncthreads();
                if (lead) {
                    st->after_ref = afterRef;
                    st->after_num = afterNum;
                    st->count = cnt + 1; // count++ is evaluated whenever the first operand is true ... (see note)
                }
                __syncthreads();
                if (cond) {
                    if (st->num_cam < sc.cfg.minCamNum) { // :142-147
                        __syncthreads();
                        if (lead) {
                            st->fitness = DBL_MAX;
                            st->priority = DBL_MAX;
                            st->dropped = 1;
                        }
                        __syncthreads();
                    } else {
                        again = true;
                        __syncthreads();
                        if (lead) {
                            st->before_ref = st->ref_cam;
                            st->before_num = st->num_cam;
                            st->stage = PAIS_STAGE_PSO;
                        }
                        __syncthreads();
                    }
                }
            }
            if (!again) {
                set_priority_and_image_point(sc, st, lane);
                __syncthreads();
                // mvs.cpp:215 / :574 follows.  Centre and normal are those of the first call; with the same reference camera
                // and LOD the homographies -- and so the region ratios of the cameras that stayed -- are too
                if (lead)
                    st->stage = st->dropped ? PAIS_STAGE_DONE
                                            : ((st->ref_cam == refBefore && st->lod == lodBefore)
                                                   ? (removedNone ? PAIS_STAGE_AFTER2_SAME : PAIS_STAGE_AFTER2_KEEP)
                                                   : PAIS_STAGE_AFTER2);
            }
            __syncthreads();
            if (lead && again) atomicAdd(&counters[1], 1);
        } else {
            // mvs.cpp:215 / :574.  PAIS_STAGE_AFTER2_SAME: the first call removed nothing and the setters left the reference
            // camera and the LOD alone, so this call would warp the same patches of the same cameras with the same
            // homographies, build the same table and again remove nothing -- only its call count is carried
            if (stg == PAIS_STAGE_AFTER2_SAME) {
                if (lead) st->ncc_tables += 1;
            } else {
                remove_invisible_camera(sc, st, hp, table, Hn, flag, rat, lane, wave);
            }
            __syncthreads();
            // the scene half of MVS::runtimeFiltering (mvs.cpp:851-863) for the caller's insertPatch: one thread per camera
            int finalStage = PAIS_DONE;
            if (!st->dropped) {
                bool off = false;
                const double X[3] = {st->center[0], st->center[1], st->center[2]};
                for (int i = threadIdx.x; i < sc.numCams; i += 64 * AFTER_WAVES) {
                    const DevCamera &cam = sc.cams[i];
                    double pt[2];
                    project_raw(cam.R, cam.T, cam.focal, cam.pp, 1.0, X, pt);
                    if (!in_image_d(pt, cam.w[0], cam.h[0])) {
                        off = true;
                    } else {
                        // (the reference reads at(cvRound(y), cvRound(x)); defined as the edge pixel within half a pixel of the
                        // right / bottom edge, as in the host statement)
                        const int rx = min(cv_round(pt[0]), cam.w[0] - 1), ry = min(cv_round(pt[1]), cam.h[0] - 1);
                        if (sc.imgBlob[cam.imgOff[0] + (size_t)ry * cam.w[0] + rx] == 0) off = true;
                    }
                }
                finalStage = __syncthreads_or(off ? 1 : 0) ? PAIS_DONE_OFF_SCENE : PAIS_DONE_IN_SCENE;
            }
            if (lead) st->stage = finalStage;
        }
        __syncthreads();
        if (lead) {
            atomicAdd(&stat[3], (unsigned long long)(st->ncc_tables - nccBefore));
            atomicAdd(&stat[4], (unsigned long long)(st->ncc_tables - nccBefore) * (unsigned long long)st->total_cam_num);
        }
        __syncthreads();
        copy_record(&recs[c], st, threadIdx.x, 64 * AFTER_WAVES);
    }
}

// ------------------------------------------------------------ record wire ---
// include/pais_hip.h "wire format of a record": one thread per 4-byte word of a slot; the layout below is the one
// pais_pack_records / pais_unpack_records (pais_capi.hip) state on the host
__global__ __launch_bounds__(256) void k_pack_records(const pais_patch_result *recs, int n, int Kw, uint32_t *wire)
{
    const int wordsPerSlot = 52 + 5 * Kw; // 208 / 4 + Kw * (16 + 4) / 4
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < (size_t)n * wordsPerSlot; t += (size_t)gridDim.x * 256) {
        const int r = (int)(t / wordsPerSlot), w = (int)(t - (size_t)r * wordsPerSlot);
        const uint32_t *src = (const uint32_t *)&recs[r];
        // record words: [0,34) the 17 doubles; [34,290) imgPoint; [290,292) key; [292,300) type..pso_evals; [300,364) cam_idx;
        // [364,372) stage..ncc_tables
        int sw;
        if (w < 34) sw = w;
        else if (w < 36) sw = 290 + (w - 34);
        else if (w < 44) sw = 292 + (w - 36);
        else if (w < 52) sw = 364 + (w - 44);
        else if (w < 52 + 4 * Kw) sw = 34 + (w - 52);
        else sw = 300 + (w - 52 - 4 * Kw);
        wire[t] = src[sw];
    }
}

// ------------------------------------------------------- scene preparation ---
// float2 tap copy of the byte blob (pais_internal.h PaisImgT): one thread per pixel, coalesced
// header of a rank's block in a sharded batch's exchange (include/pais_hip.h pais_wire_header): the ring's own words decide
// whether the pass completed, so that the status travels with the records without a host round trip before the collective
__global__ __launch_bounds__(64) void k_wire_header(uint32_t *hdr, const RingCtl *ctl, int ringN, int rank, int count, int hostRc, uint32_t userWord)
{
    if (threadIdx.x >= 16) return;
    int rc = hostRc;
    if (rc == 0 && ctl) {
        unsigned done = 0, err = 0;
        for (int r = 0; r < PAIS_RINGS; ++r) { done += ctl[r].done; err |= ctl[r].error; }
        if (err != 0 || done != (unsigned)ringN) rc = PAIS_WIRE_RC_RING_RETRY;
    }
    const uint32_t w[4] = {PAIS_WIRE_MAGIC, (uint32_t)rc, (uint32_t)count, (uint32_t)rank};
    hdr[threadIdx.x] = threadIdx.x < 4 ? w[threadIdx.x] : (threadIdx.x == 4 ? userWord : 0u);
}

__global__ __launch_bounds__(256) void k_expand_image(const uint8_t *img, PaisImgT *out, size_t n)
{
    // grid-stride: a blob of several GB has more pixels than one launch may have threads (2^32)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int a = img[i], b = (i + 1 < n) ? img[i + 1] : 0; // the last column's difference is never tapped
        PaisImgT v;
        v.x = a;
        v.y = b - a;
        out[i] = v;
    }
}
// minimum / maximum Sobel magnitude of one level (the statements of k_sobel_mag in pais_pyramid.hip without the map):
// magnitudes are >= 0, so their bit patterns order like the values; one atomic pair per workgroup
__global__ __launch_bounds__(256) void k_level_edge_minmax(const uint8_t *img, int w, int h, unsigned long long *minmax)
{
    __shared__ unsigned long long red[8];
    const int x = blockIdx.x * 256 + threadIdx.x;
    unsigned long long lo = ~0ULL, hi = 0ULL;
    if (x < w) {
        const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
        for (int r = 0; r < 16; ++r) {
            const int y = blockIdx.y * 16 + r;
            if (y >= h) break;
            const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
            const double gx = (double)img[(size_t)y * w + xr] - (double)img[(size_t)y * w + xl];
            const double gy = (double)img[(size_t)yd * w + x] - (double)img[(size_t)yu * w + x];
            const unsigned long long u = (unsigned long long)__double_as_longlong(sqrt(gx * gx + gy * gy));
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

// -------------------------------------------------------- k_neighbor_count ---
// MVS::neighborPatchFiltering (mvs.cpp:448-524), the O(n^2) part: for every patch the number of OTHER patches
// whose centre lies within neighborRadius (the reference sorts all distances and counts up to the first one
// beyond the radius: the same number).  One thread per patch i; the centres are streamed through LDS in tiles
// of 256.  dist = sqrt(dx*dx + dy*dy + dz*dz) accumulated in that order (cv::norm of a Vec3d), compared as
// `!(dist > radius)` like the reference's `if (dist > neighborRadius) break`.
__global__ __launch_bounds__(256) void k_neighbor_count(const double *centers, int n, double radius, int32_t *counts)
{
    __shared__ double tile[256 * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool has = i < n;
    const double cx = has ? centers[3 * i] : 0, cy = has ? centers[3 * i + 1] : 0, cz = has ? centers[3 * i + 2] : 0;
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        const int m = (n - base) < 256 ? (n - base) : 256;
        __syncthreads();
        for (int t = threadIdx.x; t < 3 * m; t += 256) tile[t] = centers[3 * (size_t)base + t];
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            const double dx = cx - tile[3 * j], dy = cy - tile[3 * j + 1], dz = cz - tile[3 * j + 2];
            double s2 = 0;
            s2 += dx * dx;
            s2 += dy * dy;
            s2 += dz * dz;
            const double dist = sqrt(s2);
            cnt += ((base + j) != i && !(dist > radius)) ? 1 : 0;
        }
    }
    if (has) counts[i] = cnt;
}

// --------------------------------------------------------------- launchers ---
namespace pais_launch {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: the largest size set so far is remembered per
// (kernel, device), so a second context on another GPU of the same process sets it again
struct LdsAttr {
    size_t setFor[64] = {0};
    hipError_t ensure(const void *fn, size_t lds)
    {
        if (lds <= 64 * 1024) return hipSuccess;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) dev = 0;
        if (lds <= setFor[dev]) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) setFor[dev] = lds;
        return e;
    }
};

static size_t after_lds_bytes(int Kmax)
{
    size_t off = (sizeof(pais_patch_result) + 15) & ~(size_t)15;
    off += sizeof(double) * Kmax * Kmax;
    off += sizeof(double) * 9 * Kmax;
    off += 16; // the drop flag of remove_invisible_camera
    return off;
}
// two window pixels per lane only while the LDS scratch leaves >= 3 waves per SIMD (M = K - 1 cameras are tapped)
// workgroups of an evaluation launch over `tasks` waves (grid-stride beyond 262144 workgroups)
static inline int eval_grid(long tasks)
{
    long g = (tasks + PAIS_WG_WAVES - 1) / PAIS_WG_WAVES;
    if (g > 262144) g = 262144;
#if PAIS_XCD_SWIZZLE
    g = (g + 7) & ~7L;
#endif
    return (int)(g < 1 ? 1 : g);
}
// shape of the evaluation kernels for a batch (pais_eval.hpp): 0: NS 2 / LDS accumulators, 1: NS 2 / register
// accumulators, 2: NS 1 / register accumulators
#ifndef PAIS_TWO_PIXELS_REG_MAXK
#define PAIS_TWO_PIXELS_REG_MAXK 12
#endif
static inline int eval_shape(int Kmax) { return Kmax <= PAIS_TWO_PIXELS_MAXK ? 0 : (Kmax <= PAIS_TWO_PIXELS_REG_MAXK ? 1 : 2); }
// calls F<NS, BYTES, ACCR>(args...) for the batch's shape and the scene's tap representation
#define PAIS_SHAPE_DISPATCH(F, ...)                                                                         \
    do {                                                                                                     \
        const int shape_ = eval_shape(Kmax);                                                                 \
        if (byte_taps(sc))                                                                                   \
            return shape_ == 0 ? F<2, true, false>(__VA_ARGS__)                                              \
                               : (shape_ == 1 ? F<2, true, true>(__VA_ARGS__) : F<1, true, true>(__VA_ARGS__)); \
        return shape_ == 0 ? F<2, false, false>(__VA_ARGS__)                                                 \
                           : (shape_ == 1 ? F<2, false, true>(__VA_ARGS__) : F<1, false, true>(__VA_ARGS__)); \
    } while (0)
// the scene decides what the taps read (pais_internal.h PaisImgT)
static inline bool byte_taps(const DevScene &sc) { return sc.imgF == nullptr; }

size_t eval_block_bytes_host(int Kmax) { return eval_block_bytes(Kmax); }
size_t win_bytes_per_candidate(const DevScene &sc) { return sizeof(WinPix) * (size_t)win_stride(sc); }

template <int NS, bool BYTES, bool ACCR>
static hipError_t fitness_launch(const DevScene &sc, const int32_t *idx, const double *particles, double *out, int nEvals, int Kmax,
                                 const unsigned char *evalBlocks, const void *win, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = eval_lds_bytes(NS, Kmax, ACCR) * PAIS_WG_WAVES;
    hipError_t e = attr.ensure((const void *)k_fitness<NS, BYTES, ACCR>, lds);
    if (e != hipSuccess) return e;
    const int grid = eval_grid(nEvals);
    hipLaunchKernelGGL((k_fitness<NS, BYTES, ACCR>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, idx, particles, out, nEvals, Kmax, evalBlocks,
                       eval_block_bytes(Kmax), (const WinPix *)win);
    return hipGetLastError();
}
hipError_t fitness(const DevScene &sc, const pais_patch_state *states, int nStates, const int32_t *idx, const double *particles,
                   double *out, int nEvals, int Kmax, unsigned char *evalBlocks, void *win, int literal, hipStream_t stream)
{
    if (nEvals <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_state_blocks, dim3(nStates < 65536 ? nStates : 65536), dim3(64), 0, stream, sc, states, nStates, evalBlocks,
                       eval_block_bytes(Kmax), (WinPix *)win);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (literal) { // PAIS_ARITH=literal (pais_literal.hpp)
        static LdsAttr attr;
        const size_t lds = literal_lds_bytes(Kmax);
        e = attr.ensure((const void *)k_fitness_lit, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_fitness_lit, dim3(nEvals < (1 << 20) ? nEvals : (1 << 20)), dim3(64), lds, stream, sc, idx, particles, out, nEvals, Kmax,
                           evalBlocks, eval_block_bytes(Kmax));
        return hipGetLastError();
    }
    PAIS_SHAPE_DISPATCH(fitness_launch, sc, idx, particles, out, nEvals, Kmax, evalBlocks, win, stream);
}

// k_begin + the set-up of the first PSO run of every candidate (what k_pso_init does in later passes)
hipError_t begin(const DevScene &sc, const pais_candidate *cands, pais_patch_result *recs, int n, unsigned char *states, int Nmax,
                 int *activeList, int *activeCount, unsigned char *evalBlocks, void *win, int Kmax, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    int grid = n < 16384 ? n : 16384;
    hipLaunchKernelGGL((k_begin<true>), dim3(grid), dim3(64), 0, stream, sc, cands, recs, n, states, Nmax, activeList, activeCount, evalBlocks,
                       eval_block_bytes(Kmax), (WinPix *)win);
    return hipGetLastError();
}

hipError_t expand_image(const uint8_t *img, PaisImgT *out, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_expand_image, dim3((unsigned)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, stream, img, out, n);
    return hipGetLastError();
}
// minmax[0] = ~0, minmax[1] = 0 on entry (ordered bit patterns of the minimum / maximum magnitude on exit)
hipError_t level_edge_minmax(const uint8_t *img, int w, int h, unsigned long long *minmax, hipStream_t stream)
{
    hipLaunchKernelGGL(k_level_edge_minmax, dim3((w + 255) / 256, (h + 15) / 16), dim3(256), 0, stream, img, w, h, minmax);
    return hipGetLastError();
}
hipError_t neighbor_count(const double *centers, int n, double radius, int32_t *counts, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_neighbor_count, dim3((n + 255) / 256), dim3(256), 0, stream, centers, n, radius, counts);
    return hipGetLastError();
}

hipError_t pack_records(const pais_patch_result *recs, int n, int Kw, void *wire, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const size_t words = (size_t)n * (52 + 5 * Kw);
    const size_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, recs, n, Kw, (uint32_t *)wire);
    return hipGetLastError();
}

hipError_t wire_header(void *header, const unsigned *ringCtl, int ringN, int rank, int count, int hostRc, uint32_t userWord, hipStream_t stream)
{
    hipLaunchKernelGGL(k_wire_header, dim3(1), dim3(64), 0, stream, (uint32_t *)header, (const RingCtl *)ringCtl, ringN, rank, count, hostRc, userWord);
    return hipGetLastError();
}

size_t pso_state_bytes_host(int Nmax) { return pso_state_bytes(Nmax); }
hipError_t pso_init(const DevScene &sc, const pais_patch_result *recs, int n, unsigned char *states, int Nmax, int *activeList,
                    int *activeCount, unsigned char *evalBlocks, void *win, int Kmax, hipStream_t stream)
{
    int grid = n < 65536 ? n : 65536;
    hipLaunchKernelGGL(k_pso_init, dim3(grid), dim3(64), 0, stream, sc, recs, n, states, Nmax, activeList, activeCount, evalBlocks,
                       eval_block_bytes(Kmax), (WinPix *)win);
    return hipGetLastError();
}
template <int NS, bool BYTES, bool ACCR>
static hipError_t pso_eval2_launch(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks,
                                   const void *win, int pendingOnly, unsigned long long *verify, const double *pre, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = eval_lds_bytes(NS, Kmax, ACCR) * PAIS_WG_WAVES;
    static LdsAttr attrPre;
    hipError_t e = pre ? attrPre.ensure((const void *)k_pso_eval2<NS, BYTES, ACCR, true>, lds) : attr.ensure((const void *)k_pso_eval2<NS, BYTES, ACCR, false>, lds);
    if (e != hipSuccess) return e;
    // pending-only (behind k_pso_tile): a handful of particles at most -- a few waves scan the flags (grid-stride loop);
    // one workgroup per particle would queue tens of thousands of workgroups, each asking for LDS, behind the tile
    // kernel of the other sub-stream, which holds every CU's LDS (measured: 2.2 ms per launch spent waiting)
    const int grid = pendingOnly == 1 ? (int)(((long)n * Nmax < 256) ? (long)n * Nmax : 256) : eval_grid((long)n * Nmax);
    if (pre)
        hipLaunchKernelGGL((k_pso_eval2<NS, BYTES, ACCR, true>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, states, n, Nmax, Kmax, evalBlocks,
                           eval_block_bytes(Kmax), (const WinPix *)win, pendingOnly, verify, pre);
    else
        hipLaunchKernelGGL((k_pso_eval2<NS, BYTES, ACCR, false>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, states, n, Nmax, Kmax, evalBlocks,
                           eval_block_bytes(Kmax), (const WinPix *)win, pendingOnly, verify, pre);
    return hipGetLastError();
}
// the evaluation launch of large batches: `states`, `evalBlocks`, `win` point at the slice's first candidate
hipError_t pso_eval(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, const void *win,
                    int pendingOnly, unsigned long long *verify, hipStream_t stream, const double *pre)
{
    PAIS_SHAPE_DISPATCH(pso_eval2_launch, sc, states, n, Nmax, Kmax, evalBlocks, win, pendingOnly, verify, pre, stream);
}
// the evaluation records of the initial swarm of every active candidate (pais_pre.hpp); behind k_begin / k_pso_init
size_t pre_bytes_per_candidate(int Nmax, int Kmax) { return pre_rec_bytes(Kmax) * (size_t)Nmax; }
// (the one-pixel ring kernel with the step's record code needs 177 VGPRs -- 2 waves / SIMD instead of 3: it keeps setting itself up)
#ifndef PAIS_PRE_RING_NS1
#define PAIS_PRE_RING_NS1 0
#endif
bool pre_ring_ok(int Kmax) { return PAIS_PRE_RING_NS1 || eval_shape(Kmax) != 2; }
hipError_t pso_setup0(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, double *pre,
                      hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    static LdsAttr attr;
    const size_t lds = eval_block_bytes(Kmax) + sizeof(double) * 3 * (size_t)Nmax + pre_scratch_bytes(Nmax);
    hipError_t e = attr.ensure((const void *)k_pso_setup0, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pso_setup0, dim3(n < 65536 ? n : 65536), dim3(64), lds, stream, sc, states, n, Nmax, evalBlocks, eval_block_bytes(Kmax), pre, Kmax);
    return hipGetLastError();
}
// the evaluation launch of a PSO iteration under PAIS_ARITH=literal (pais_literal.hpp)
hipError_t pso_eval_literal(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = literal_lds_bytes(Kmax);
    hipError_t e = attr.ensure((const void *)k_pso_eval_lit, lds);
    if (e != hipSuccess) return e;
    const long tasks = (long)n * Nmax;
    hipLaunchKernelGGL(k_pso_eval_lit, dim3((unsigned)(tasks < (1 << 20) ? tasks : (1 << 20))), dim3(64), lds, stream, sc, states, n, Nmax, Kmax, evalBlocks,
                       eval_block_bytes(Kmax));
    return hipGetLastError();
}
// many-camera batches: the tile kernel (pais_tile.hpp) can take the evaluation launch of the large-batch pipeline
bool tile_eligible(int Kmax) { return eval_shape(Kmax) == 2 && Kmax <= TILE_MAX_CAMS; }
template <int NS, int NP>
static hipError_t pso_tile_launch(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks,
                                  const void *win, int stripSteps, unsigned long long *dbg, double *hscr, size_t hscrBytes, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = ((160 * 1024) / TILE_WGS_PER_CU) & ~(size_t)1023, fixed = tile_fixed_lds_bytes(Kmax);
    if (fixed + 4096 > lds) return hipErrorInvalidValue;
    hipError_t e = attr.ensure((const void *)k_pso_tile<NS, NP>, lds);
    if (e != hipSuccess) return e;
    const int groups = (Nmax + TILE_WAVES - 1) / TILE_WAVES;
    long grid = (long)n * groups;
    if (grid > 65536) grid = 65536;
    // one slot of the homography scratch per wave of the grid (pais_tile.hpp PAIS_TILE_SCALAR_H); a workgroup takes the whole LDS
    // of a CU, so a grid of a few workgroups per CU (grid-stride over the tasks) loses nothing
    if (PAIS_TILE_SCALAR_H) {
        const long slots = (long)(hscrBytes / (sizeof(double) * PAIS_H_STRIDE * (size_t)Kmax * TILE_WAVES));
        if (slots < 1) return hipErrorInvalidValue;
        if (grid > slots) grid = slots;
    }
    hipLaunchKernelGGL((k_pso_tile<NS, NP>), dim3((unsigned)grid), dim3(64 * TILE_WAVES), lds, stream, sc, states, n, Nmax, Kmax, evalBlocks,
                       eval_block_bytes(Kmax), (const WinPix *)win, getenv("PAIS_TILE_NOTILES") ? 0 : (int)(lds - fixed), groups, stripSteps, dbg,
                       hscr);
    return hipGetLastError();
}
// the split kernel (pais_tile2.hpp): sixteen waves, the cameras of a particle shared by two of them
template <int NP>
static hipError_t pso_tile2_launch(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks,
                                   const void *win, int stripSteps, int bias, unsigned long long *dbg, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = (160 * 1024) & ~(size_t)1023, fixed = tile2_fixed_lds_bytes(Kmax);
    if (fixed + 4096 > lds) return hipErrorInvalidValue;
    hipError_t e = attr.ensure((const void *)k_pso_tile2<NP>, lds);
    if (e != hipSuccess) return e;
    const int groups = (Nmax + TILE2_SLOTS - 1) / TILE2_SLOTS;
    long grid = (long)n * groups;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL((k_pso_tile2<NP>), dim3((unsigned)grid), dim3(64 * TILE2_WAVES), lds, stream, sc, states, n, Nmax, Kmax, evalBlocks,
                       eval_block_bytes(Kmax), (const WinPix *)win, getenv("PAIS_TILE_NOTILES") ? 0 : (int)(lds - fixed), groups, stripSteps, bias, dbg);
    return hipGetLastError();
}
hipError_t pso_tile(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, const void *win,
                    int strip2, int strip1, int forceNs1, int split, int stripSplit, int bias, unsigned long long *dbg, double *hscr,
                    size_t hscrBytes, hipStream_t stream)
{
    if (split && Kmax > split - 1) { // `split` - 1 = camera count above which a batch takes the split kernel (1: every batch)
        // the colours of at most NP pairs per wave: Kmax <= 4 NP cameras, with room for the first half's larger share
        if (Kmax <= 28) return pso_tile2_launch<8>(sc, states, n, Nmax, Kmax, evalBlocks, win, stripSplit, bias, dbg, stream);
        if (Kmax <= 44) return pso_tile2_launch<12>(sc, states, n, Nmax, Kmax, evalBlocks, win, stripSplit, bias, dbg, stream);
        return pso_tile2_launch<16>(sc, states, n, Nmax, Kmax, evalBlocks, win, stripSplit, bias, dbg, stream);
    }
    // two pixels per lane while the colours of 2 x 32 cameras fit the registers; one pixel per lane beyond.  Strip lengths
    // swept on the full-size dome (profiles/r03_dome_tile_sweep.txt): 14 / 24 steps; longer strips = fewer barriers, until the
    // tiles of a strip stop fitting the tile area (cameras then tap global memory)
    if (Kmax <= 32 && !forceNs1) return pso_tile_launch<2, 16>(sc, states, n, Nmax, Kmax, evalBlocks, win, (strip2 + 1) & ~1, dbg, hscr, hscrBytes, stream);
    return pso_tile_launch<1, 32>(sc, states, n, Nmax, Kmax, evalBlocks, win, strip1, dbg, hscr, hscrBytes, stream);
}
template <int P, int NS, bool BYTES, bool ACCR>
static hipError_t pso_iter_launch(const DevScene &sc, unsigned char *states, const int *activeList, const int *activeCount,
                                  int listLo, int listHi, int Nmax, int Kmax, pais_patch_result *recs, unsigned long long *stat, int L,
                                  int finishOnly, const unsigned char *evalBlocks, const void *win, hipStream_t stream)
{
    static LdsAttr attr;
    const size_t lds = eval_lds_bytes(NS, Kmax, ACCR) * PAIS_WG_WAVES;
    hipError_t e = attr.ensure((const void *)k_pso_iter<P, NS, BYTES, ACCR>, lds);
    if (e != hipSuccess) return e;
    const int grid = eval_grid((long)(listHi - listLo) * (finishOnly ? 1 : Nmax * P));
    hipLaunchKernelGGL((k_pso_iter<P, NS, BYTES, ACCR>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, states, activeList, activeCount, listLo,
                       listHi, Nmax, Kmax, recs, stat, L, finishOnly, evalBlocks, eval_block_bytes(Kmax), (const WinPix *)win);
    return hipGetLastError();
}
#define PAIS_ITER_WRAPPER(P)                                                                  \
    template <int NS, bool BYTES, bool ACCR, class... A> static hipError_t pso_iter_launch##P(const A &...a) \
    {                                                                                         \
        return pso_iter_launch<P, NS, BYTES, ACCR>(a...);                                     \
    }
PAIS_ITER_WRAPPER(1)
PAIS_ITER_WRAPPER(2)
PAIS_ITER_WRAPPER(4)
#undef PAIS_ITER_WRAPPER
hipError_t pso_iter(const DevScene &sc, unsigned char *states, const int *activeList, const int *activeCount, int listLo,
                    int listHi, int Nmax, int Kmax, pais_patch_result *recs, unsigned long long *stat, int L, int finishOnly,
                    int nparts, const unsigned char *evalBlocks, const void *win, hipStream_t stream)
{
    if (listHi <= listLo) return hipSuccess;
#define PAIS_ARGS sc, states, activeList, activeCount, listLo, listHi, Nmax, Kmax, recs, stat, L, finishOnly, evalBlocks, win, stream
    if (nparts == 4) { PAIS_SHAPE_DISPATCH(pso_iter_launch4, PAIS_ARGS); }
    if (nparts == 2) { PAIS_SHAPE_DISPATCH(pso_iter_launch2, PAIS_ARGS); }
    PAIS_SHAPE_DISPATCH(pso_iter_launch1, PAIS_ARGS);
#undef PAIS_ARGS
}
// k_pso_ring: a whole PSO pass of a large batch in one launch (ring: n * Nmax * (maxIt + 2) words, ctl: 8 words, arrive: n ints --
// all zeroed / emptied here).  waves: resident waves to work the ring off with.
static size_t ring_seg_words(int n, int Nmax, int maxIt) { return (size_t)((n + PAIS_RINGS - 1) / PAIS_RINGS) * Nmax * ((size_t)maxIt + 2); }
size_t ring_words(int n, int Nmax, int maxIt) { return ring_seg_words(n, Nmax, maxIt) * PAIS_RINGS; }
template <int NS, bool BYTES, bool ACCR>
static hipError_t pso_ring_launch(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax, int Kmax, int maxIt,
                                  const unsigned char *evalBlocks, const void *win, unsigned *ring, unsigned *ctl, int *arrive,
                                  unsigned long long *stat, int waves, int phase, unsigned long long timeoutTicks, double *pre, hipStream_t stream)
{
    static LdsAttr attr;
    size_t per = eval_lds_bytes(NS, Kmax, ACCR);
    // (with records the step's scratch lies behind the evaluation block and includes the scratch of swarm_eval_setup)
    const size_t stepBytes = pre ? eval_block_bytes(Kmax) + step_lds_bytes(Nmax) : sizeof(double) * (size_t)Nmax * (3 * 4 + 2) + 16;
    if (per < stepBytes) per = stepBytes;
    per = (per + 15) & ~(size_t)15;
    const size_t lds = per * PAIS_WG_WAVES;
    static LdsAttr attrPre;
    hipError_t e = pre ? attrPre.ensure((const void *)k_pso_ring<NS, BYTES, ACCR, true>, lds) : attr.ensure((const void *)k_pso_ring<NS, BYTES, ACCR, false>, lds);
    if (e != hipSuccess) return e;
    const size_t words = ring_words(n, Nmax, maxIt), seg = ring_seg_words(n, Nmax, maxIt);
    if (seg >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    if (phase == 0) { // empty rings, zero counters, the tasks of iteration 0
        e = hipMemsetAsync(ring, 0xFF, words * sizeof(unsigned), stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(ctl, 0, sizeof(RingCtl) * PAIS_RINGS, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(arrive, 0, sizeof(int) * PAIS_ARRIVE_STRIDE * (size_t)n, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ring_init, dim3((n + 255) / 256), dim3(256), 0, stream, states, n, Nmax, ring, (unsigned)seg, (RingCtl *)ctl);
        return hipGetLastError();
    }
    long tasks = (long)n * Nmax;
    waves *= (NS == 1 ? 4 * PAIS_NS1_WAVES : 4 * PAIS_NS2_WAVES); // `waves` arrives as the number of CUs: resident waves per CU by shape
    int grid = (int)(tasks < waves ? tasks : waves);
    grid = (grid + PAIS_RINGS - 1) / PAIS_RINGS * PAIS_RINGS;
    if (pre)
        hipLaunchKernelGGL((k_pso_ring<NS, BYTES, ACCR, true>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, recs, states, n, Nmax, Kmax, evalBlocks,
                           eval_block_bytes(Kmax), (const WinPix *)win, ring, (unsigned)seg, (RingCtl *)ctl, arrive, stat, per, timeoutTicks, pre);
    else
        hipLaunchKernelGGL((k_pso_ring<NS, BYTES, ACCR, false>), dim3(grid), dim3(64 * PAIS_WG_WAVES), lds, stream, sc, recs, states, n, Nmax, Kmax, evalBlocks,
                           eval_block_bytes(Kmax), (const WinPix *)win, ring, (unsigned)seg, (RingCtl *)ctl, arrive, stat, per, timeoutTicks, pre);
    return hipGetLastError();
}
hipError_t pso_ring(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax, int Kmax, int maxIt,
                    const unsigned char *evalBlocks, const void *win, unsigned *ring, unsigned *ctl, int *arrive, unsigned long long *stat,
                    int waves, int phase, unsigned long long timeoutTicks, hipStream_t stream, double *pre)
{
    PAIS_SHAPE_DISPATCH(pso_ring_launch, sc, recs, states, n, Nmax, Kmax, maxIt, evalBlocks, win, ring, ctl, arrive, stat, waves, phase, timeoutTicks, pre, stream);
}
hipError_t pso_step(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax,
                    unsigned long long *stat, hipStream_t stream, const unsigned char *evalBlocks, double *pre, int Kmax)
{
    int grid = n < 65536 ? n : 65536;
    static LdsAttr attr;
    const size_t lds = pre ? eval_block_bytes(Kmax) + step_lds_bytes(Nmax) : sizeof(double) * (size_t)Nmax * (3 * 4 + 2);
    hipError_t e = attr.ensure((const void *)k_pso_step, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pso_step, dim3(grid), dim3(64), lds, stream, sc, recs, states, n, Nmax, stat, evalBlocks, eval_block_bytes(Kmax), pre, Kmax);
    return hipGetLastError();
}

hipError_t after(const DevScene &sc, pais_patch_result *recs, int n, double *hpScratch, int grid, int *counters,
                 unsigned long long *stat, int Kmax, double *ratios, int *nextCounters, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    static LdsAttr attr1, attr2;
    size_t lds = after_lds_bytes(Kmax);
    const size_t hpBytes = sizeof(double) * (size_t)Kmax * sc.cfg.patchSize * sc.cfg.patchSize;
    const int hpInLds = hpBytes <= 60 * 1024 ? 1 : 0; // <= 2 workgroups per CU otherwise
    if (hpInLds) lds += hpBytes;
    hipError_t e = attr1.ensure((const void *)k_after<1>, lds);
    if (e != hipSuccess) return e;
    e = attr2.ensure((const void *)k_after<2>, lds);
    if (e != hipSuccess) return e;
    const int rgrid = (int)(((long)n * Kmax + 63) / 64);
    hipLaunchKernelGGL(k_region_ratio, dim3(rgrid), dim3(64), 0, stream, sc, recs, n, PAIS_STAGE_AFTER, ratios, Kmax);
    hipLaunchKernelGGL((k_after<1>), dim3(grid), dim3(64 * AFTER_WAVES), lds, stream, sc, recs, n, hpScratch, counters, stat, Kmax, ratios, hpInLds,
                       (int *)nullptr);
    hipLaunchKernelGGL(k_region_ratio, dim3(rgrid), dim3(64), 0, stream, sc, recs, n, PAIS_STAGE_AFTER2, ratios, Kmax);
    hipLaunchKernelGGL((k_after<2>), dim3(grid), dim3(64 * AFTER_WAVES), lds, stream, sc, recs, n, hpScratch, counters, stat, Kmax, ratios, hpInLds,
                       nextCounters);
    return hipGetLastError();
}

void ring_profile_print()
{
#if PAIS_RING_PROFILE
    unsigned long long h[8] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ringProf), sizeof(h)) != hipSuccess || h[6] == 0) return;
    const char *nm[6] = {"index (head atomic)", "entry wait", "loads + staging", "evaluation", "fitness store + arrival", "swarm step + publish"};
    double tot = 0;
    for (int k = 0; k < 6; ++k) tot += (double)h[k];
    fprintf(stderr, "[ring profile] %llu tasks, %llu steps; s_memtime ticks per task:\n", h[6], h[7]);
    for (int k = 0; k < 6; ++k)
        fprintf(stderr, "[ring profile]   %-26s %10.1f  (%5.1f %%)%s\n", nm[k], (double)h[k] / (double)h[6], 100.0 * (double)h[k] / tot,
                k == 5 ? "  (per step: see steps)" : "");
    fprintf(stderr, "[ring profile]   per step %.1f ticks\n", h[7] ? (double)h[5] / (double)h[7] : 0.0);
#endif
}
} // namespace pais_launch
