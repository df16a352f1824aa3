// pais_internal.h -- device-side scene layout shared by pais_kernels.hip and pais_capi.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pais_hip.h"

// refine() progress of a record (pais_patch_result::stage)
#define PAIS_STAGE_DONE  0
#define PAIS_STAGE_PSO   1  /* psoOptimization() pending (patch.cpp:153)           */
#define PAIS_STAGE_AFTER 2  /* PSO finished, removeInvisibleCamera etc. pending     */
#define PAIS_STAGE_AFTER2 3 /* refine() has ended: the caller's removeInvisibleCamera (mvs.cpp:215 / :574) pending */
#define PAIS_STAGE_AFTER2_KEEP 4 /* ... and the region ratios of the first call are still valid                     */
#define PAIS_STAGE_AFTER2_SAME 7 /* ... and the first call removed no camera: the second one sees the very same inputs (5, 6: PAIS_DONE_*) */

// HBM layout: one DevCamera per camera in a dense array;
// every pyramid level of every camera repacked row-major with stride == width
// into one byte blob (levels 256-byte aligned), edge levels likewise as doubles.
// The cost kernel's bilinear taps read either a float2 copy of the byte blob (imgF) or the byte blob itself
// (DevScene::imgF == nullptr): see PaisImgT below.
struct DevCamera {
    double KR[9], KT[3], R[9], T[3], C[3], optN[3], focal[2], pp[2];
    int maxLOD;
    int pad;
    int w[PAIS_MAX_LEVELS], h[PAIS_MAX_LEVELS];
    // offsets into DevScene::imgBlob (bytes) / DevScene::edgeBlob (doubles): keeping the base in a
    // kernel argument lets the compiler emit global_load (saddr) instead of flat_load for every tap
    uint64_t imgOff[PAIS_MAX_LEVELS];
    uint64_t edgeOff[PAIS_MAX_LEVELS];
    // edge maps evaluated on the fly (DevScene::edgeBlob == nullptr while adaptiveGradientEnable is set): the minimum and
    // maximum Sobel magnitude of every level, what the per-level normalisation of camera.cpp:72-77,87-91 needs
    double edgeMin[PAIS_MAX_LEVELS], edgeMax[PAIS_MAX_LEVELS];
};

// What the cost taps read (a property of the uploaded scene; the sampled values are identical either way):
//   float2 {I(x), I(x+1) - I(x)}  a copy of the byte blob with identical element offsets (imgOff): a tap row = one aligned
//                                 8-byte load, no unpacking, no subtraction.  +10 % evaluations/s where the working set
//                                 of a batch sits in L2 anyway (pawn: 107 vs 97 M evaluations/s) -- at 8x the bytes;
//   the byte blob itself          a tap row = one 2-byte load + byte->double conversions and one subtraction.  An eighth
//                                 of the footprint: +21 % patches/s on the 128 x 12.6 MP dome (4.5 GB instead of 36 GB
//                                 of pyramids, and the taps of a batch stop missing L2 / the TLB); equal on the ring.
// pais_create expands the float2 copy when the byte pyramids are at most PAIS_TAP_FLOAT_MAX_MB (default 256 MB, i.e. a
// 2 GB copy); larger scenes run the BYTES instantiations of the evaluation kernels.
#ifndef PAIS_TAP_DOUBLE
#define PAIS_TAP_DOUBLE 0
#endif
#if PAIS_TAP_DOUBLE
typedef double2 PaisImgT; // experiment: no conversion at the tap, twice the bytes again
#else
typedef float2 PaisImgT;
#endif

struct DevScene {
    pais_config cfg;
    const DevCamera *cams;
    const uint8_t *imgBlob;
    const PaisImgT *imgF;   // tap copy of imgBlob, same element offsets (imgOff); nullptr: the taps read imgBlob (BYTES kernels)
    const double *edgeBlob; // normalised Sobel magnitude per level as the caller built it; nullptr: evaluated on the fly
    const double *gauss; // patchDistWeight, S*S, indexed [x*S + y] (mvs.cpp:104-109)
    double lodScale[PAIS_MAX_LEVELS]; // pow(lodRatio, LOD) (camera.cpp:157, patch.cpp:309)
    uint64_t seed;
    int numCams;
    int pad;
};

// where the LDS-tile kernel's walk reads a wave's homographies from (pais_tile.hpp): 0 LDS (default), 1 / 2 the scalar cache
#ifndef PAIS_TILE_SCALAR_H
#define PAIS_TILE_SCALAR_H 0
#endif

// control words of one task ring of k_pso_ring (pais_kernels.hip RingCtl): head | tail, done, total, error
#ifndef PAIS_RINGS
#define PAIS_RINGS 8 // (16 and 32 rings -- two / four per XCD -- measured: no gain, profiles/r04_ring_count_ab.txt)
#endif
// a candidate's arrival counter (one atomic per delivered evaluation) on a cache line of its own: neighbouring candidates belong
// to other rings, i.e. other XCDs
#ifndef PAIS_ARRIVE_STRIDE
#define PAIS_ARRIVE_STRIDE 16
#endif
#define PAIS_RING_CTL_BYTES 128
#define PAIS_RING_CTL_DONE_WORD 17
#define PAIS_RING_CTL_ERROR_WORD 19

namespace pais_launch {
// evaluation block of a PSO run (pais_eval.hpp): EvalPatch + EvalCam[Kmax] bytes per candidate, and the reference window
size_t eval_block_bytes_host(int Kmax);
size_t win_bytes_per_candidate(const DevScene &sc);
hipError_t fitness(const DevScene &sc, const pais_patch_state *states, int nStates, const int32_t *idx, const double *particles,
                   double *out, int nEvals, int Kmax, unsigned char *evalBlocks, void *win, int literal, hipStream_t stream);
hipError_t pso_eval_literal(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, hipStream_t stream);
hipError_t begin(const DevScene &sc, const pais_candidate *cands, pais_patch_result *recs, int n, unsigned char *states, int Nmax,
                 int *activeList, int *activeCount, unsigned char *evalBlocks, void *win, int Kmax, hipStream_t stream);
hipError_t neighbor_count(const double *centers, int n, double radius, int32_t *counts, hipStream_t stream);
hipError_t expand_image(const uint8_t *img, PaisImgT *out, size_t n, hipStream_t stream);
hipError_t level_edge_minmax(const uint8_t *img, int w, int h, unsigned long long *minmax, hipStream_t stream);
hipError_t pack_records(const pais_patch_result *recs, int n, int Kw, void *wire, hipStream_t stream);
hipError_t wire_header(void *header, const unsigned *ringCtl, int ringN, int rank, int count, int hostRc, uint32_t userWord, hipStream_t stream);
size_t pso_state_bytes_host(int Nmax);
hipError_t pso_init(const DevScene &sc, const pais_patch_result *recs, int n, unsigned char *states, int Nmax, int *activeList,
                    int *activeCount, unsigned char *evalBlocks, void *win, int Kmax, hipStream_t stream);
hipError_t pso_eval(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, const void *win,
                    int pendingOnly, unsigned long long *verify, hipStream_t stream, const double *pre = nullptr);
                    // pre: evaluation records of the slice's first candidate (pais_pre.hpp); nullptr: the evaluation sets itself up
size_t pre_bytes_per_candidate(int Nmax, int Kmax);
bool pre_ring_ok(int Kmax);
void ring_profile_print(); // (measurement builds: -DPAIS_RING_PROFILE=1)
hipError_t pso_setup0(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, double *pre,
                      hipStream_t stream);
bool tile_eligible(int Kmax);
hipError_t pso_tile(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, const void *win,
                    int strip2, int strip1, int forceNs1, int split, int stripSplit, int bias, unsigned long long *dbg, double *hscr,
                    size_t hscrBytes, hipStream_t stream);
                    // split: the sixteen-wave kernel of pais_tile2.hpp (strips of stripSplit steps, the first half's share `bias` cameras larger)
                    // hscr: per-launch scratch for the waves' homographies (pais_tile.hpp PAIS_TILE_SCALAR_H), hscrBytes of it
hipError_t pso_iter(const DevScene &sc, unsigned char *states, const int *activeList, const int *activeCount, int listLo,
                    int listHi, int Nmax, int Kmax, pais_patch_result *recs, unsigned long long *stat, int L, int finishOnly,
                    int nparts, const unsigned char *evalBlocks, const void *win, hipStream_t stream);
hipError_t pso_step(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax,
                    unsigned long long *stat, hipStream_t stream, const unsigned char *evalBlocks = nullptr, double *pre = nullptr, int Kmax = 1);
size_t ring_words(int n, int Nmax, int maxIt);
hipError_t pso_ring(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax, int Kmax, int maxIt,
                    const unsigned char *evalBlocks, const void *win, unsigned *ring, unsigned *ctl, int *arrive, unsigned long long *stat,
                    int waves, int phase, unsigned long long timeoutTicks, hipStream_t stream, double *pre = nullptr); // phase 0: rings and counters prepared; 1: the launch
                    // timeoutTicks: longest wait of a wave for a ring entry, in ticks of the 100 MHz s_memrealtime counter
hipError_t after(const DevScene &sc, pais_patch_result *recs, int n, double *hpScratch, int grid, int *counters,
                 unsigned long long *stat, int Kmax, double *ratios, int *nextCounters, hipStream_t stream);
} // namespace pais_launch
