// pais_internal.h -- device-side scene layout shared by pais_kernels.hip and pais_capi.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pais_hip.h"

// refine() progress of a record (pais_patch_result::stage)
#define PAIS_STAGE_DONE  0
#define PAIS_STAGE_PSO   1  /* psoOptimization() pending (patch.cpp:153)           */
#define PAIS_STAGE_AFTER 2  /* PSO finished, removeInvisibleCamera etc. pending     */

// HBM layout: one DevCamera per camera in a dense array;
// every pyramid level of every camera repacked row-major with stride == width
// into one byte blob (levels 256-byte aligned), edge levels likewise as doubles.
// The cost kernel's bilinear taps read a float copy of the byte blob (imgF): two adjacent pixels come
// back with one 8-byte load and need no unpacking (4 fewer VALU instructions per tap).
struct DevCamera {
    double KR[9], KT[3], R[9], T[3], C[3], optN[3], focal[2], pp[2];
    int maxLOD;
    int pad;
    int w[PAIS_MAX_LEVELS], h[PAIS_MAX_LEVELS];
    // offsets into DevScene::imgBlob (bytes) / DevScene::edgeBlob (doubles): keeping the base in a
    // kernel argument lets the compiler emit global_load (saddr) instead of flat_load for every tap
    uint64_t imgOff[PAIS_MAX_LEVELS];
    uint64_t edgeOff[PAIS_MAX_LEVELS];
};

struct DevScene {
    pais_config cfg;
    const DevCamera *cams;
    const uint8_t *imgBlob;
    const float *imgF;   // the same pixels as floats, same element offsets (imgOff): what the cost taps read
    const double *edgeBlob;
    const double *gauss; // patchDistWeight, S*S, indexed [x*S + y] (mvs.cpp:104-109)
    double lodScale[PAIS_MAX_LEVELS]; // pow(lodRatio, LOD) (camera.cpp:157, patch.cpp:309)
    uint64_t seed;
    int numCams;
    int pad;
};

namespace pais_launch {
// evaluation block of a PSO run (pais_eval.hpp): EvalPatch + EvalCam[Kmax] bytes per candidate, and the reference window
size_t eval_block_bytes_host(int Kmax);
size_t win_bytes_per_candidate(const DevScene &sc);
hipError_t fitness(const DevScene &sc, const pais_patch_state *states, int nStates, const int32_t *idx, const double *particles,
                   double *out, int nEvals, int Kmax, unsigned char *evalBlocks, void *win, hipStream_t stream);
hipError_t begin(const DevScene &sc, const pais_candidate *cands, pais_patch_result *recs, int n, hipStream_t stream);
hipError_t neighbor_count(const double *centers, int n, double radius, int32_t *counts, hipStream_t stream);
size_t pso_state_bytes_host(int Nmax);
hipError_t pso_init(const DevScene &sc, const pais_patch_result *recs, int n, unsigned char *states, int Nmax, int *activeList,
                    int *activeCount, unsigned char *evalBlocks, void *win, int Kmax, hipStream_t stream);
hipError_t pso_eval(const DevScene &sc, unsigned char *states, int n, int Nmax, int Kmax, const unsigned char *evalBlocks, const void *win,
                    hipStream_t stream);
hipError_t pso_iter(const DevScene &sc, unsigned char *states, const int *activeList, const int *activeCount, int listLo,
                    int listHi, int Nmax, int Kmax, pais_patch_result *recs, unsigned long long *stat, int L, int finishOnly,
                    int nparts, const unsigned char *evalBlocks, const void *win, hipStream_t stream);
hipError_t pso_step(const DevScene &sc, pais_patch_result *recs, unsigned char *states, int n, int Nmax,
                    unsigned long long *stat, hipStream_t stream);
hipError_t after(const DevScene &sc, pais_patch_result *recs, int n, double *hpScratch, int grid, int *counters,
                 unsigned long long *stat, int Kmax, hipStream_t stream);
} // namespace pais_launch
