/*
 * pais_oracle.h -- CPU restatement of the pais-mvs refine/expansion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or executed from the product (pais_mvs_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * there only as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/TMVS/).  Plain C99, double precision throughout, same
 * evaluation order, same tie-breaks as the reference.
 *
 * PARITY PINNING STATUS
 *   - PSO solver (pso/psosolver.cpp, pso/particle.cpp): PINNED.  The
 *     reference's own two source files compile unmodified with g++ and are
 *     built into oracle/_ref/libpso_ref.so (oracle/Makefile); the golden
 *     traces in tests/golden/pso_*.json were produced by that library and the
 *     restatement in po_pso_run() reproduces them bit-for-bit.
 *   - Everything that touches OpenCV types (mvs/patch.cpp, mvs/mvs.cpp,
 *     mvs/camera.cpp): PARITY UNPINNED.  The reference needs OpenCV 2.4.2
 *     (TMVS.vcxproj:148) which is not in this image and may not be stood in
 *     for, the reference ships no tests/golden vectors,
 *     so these functions restate the source line by line plus the published
 *     OpenCV 2.4 algorithms they call (cv::invert 3x3, cv::fitEllipse,
 *     cvRound, gemm summation order) and are anchored on the reference's own
 *     call sites only.
 */
#ifndef PAIS_ORACLE_H
#define PAIS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_MAX_LEVELS 16   /* LOD 0..15 (config maxLOD default 15, TMVS.cpp:42) */
#define PO_MAX_VIS    64   /* max visible cameras tracked per patch             */
#define PO_MAX_PATCH_SIZE 129 /* largest window side (2 r + 1) the control variants of the literal cost handle */

#define PO_TYPE_SEED   0   /* patch.h:17 */
#define PO_TYPE_EXPAND 1   /* patch.h:18 */

/* mvs/mvs.h:19-72 (field meaning identical; bools widened to int) */
typedef struct po_config {
    int    cellSize;
    int    patchRadius;
    int    patchSize;
    int    minCamNum;
    double textureVariation;
    double visibleCorrelation;
    double minCorrelation;
    double maxFitness;
    double lodRatio;
    int    minLOD;
    int    maxLOD;
    int    maxCellPatchNum;
    double reduceNormalRange;
    int    adaptiveDistanceEnable;
    int    adaptiveDifferenceEnable;
    int    adaptiveGradientEnable;
    double distWeighting;
    double diffWeighting;
    double gradientWeighting;
    double neighborRadius;
    double neighborRadiusScalar;
    double minRegionRatio;
    double depthRangeScalar;
    int    particleNum;
    int    maxIteration;
    int    expansionStrategy;
} po_config;

/* mvs/camera.h:15-149 -- the data the hot path reads */
typedef struct po_camera {
    double focal[2];
    double pp[2];            /* principle point                         */
    double R[9];             /* rotation, row-major                     */
    double T[3];             /* translation = -R*C   (camera.cpp:120)   */
    double C[3];             /* centre                                  */
    double KR[9];            /* camera.cpp:123                          */
    double KT[3];            /* camera.cpp:124                          */
    double optN[3];          /* optical normal = R^T e_z (camera.cpp:132) */
    int    maxLOD;           /* camera.cpp:63-64                        */
    int    width[PO_MAX_LEVELS];
    int    height[PO_MAX_LEVELS];
    const uint8_t *img[PO_MAX_LEVELS];   /* gray pyramid, row-major, stride = width */
    const double  *edge[PO_MAX_LEVELS];  /* normalised Sobel magnitude pyramid (may be NULL if grad weighting off) */
} po_camera;

typedef struct po_scene {
    po_config  cfg;
    int        numCams;
    po_camera *cams;
    double    *gauss;                    /* patchDistWeight, S*S, mvs.cpp:97-114 */
    double     lodScale[PO_MAX_LEVELS];  /* pow(lodRatio, LOD) */
    uint64_t   seed;                     /* PSO stream seed */
    int        ompParticles;             /* 1: OpenMP over particles (reference structure) */
    /* Reproducibility switches.  Both 0 = the literal reference
     * behaviour (platform libm, sequential sums in the reference's pixel order).
     * Both 1 = the arithmetic the HIP kernels define: fdlibm exp/sin/cos
     * (po_detmath.h) and wave64-butterfly reduction trees; the GPU parity tests
     * compare against this mode bit for bit, and tests/test_oracle_modes.py
     * measures (on the CPU) how far the two modes drift apart. */
    int        detMath;
    int        treeSum;
    int        windowPerParticle;        /* kernel arithmetic, but the window origin from every particle's own centre
                                          * (patch.cpp:944-952) instead of once per run: separates the two differences
                                          * between the modes in tests/test_oracle_modes.py; the HIP path has no such mode */
    int        literalVariant;           /* CONTROL experiments on the literal arithmetic (tests/test_cloud_parity.py,
                                          * tests/golden/make_literal_control.py; 0 = the reference's statements).  Bit flags; each
                                          * is the same real-number function evaluated with a rounding difference a compiler / a
                                          * loop order could legitimately produce from the REFERENCE's own source:
                                          *  1: the final quotient as fitness * (1.0 / sumWeight) -- ONE rounding perturbed;
                                          *  2: the window's two sums accumulated y outer / x inner (same pixels, same values);
                                          *  4: the statements of patch.cpp:994-1041 as a contracting compiler emits them
                                          *     (gcc's default -ffp-contract=fast on an FMA target): fused multiply-adds in the
                                          *     homography rows, the bilinear sum and the weighted accumulation.
                                          *  6 = another loop order AND another compiler: the closest like-for-like of what the
                                          *     kernel arithmetic changes (reduction order, fused multiply-adds). */
    int        costLiteral;              /* with detMath / treeSum on: the COST (and only the cost) keeps the reference's statements and
                                          * its x-outer / y-inner sequential sums (patch.cpp:979-1041), with po_detmath.h's exp / sin /
                                          * cos -- what the HIP path computes under PAIS_ARITH=literal (pais_literal.hpp), bit for bit.
                                          * Differs from the all-literal mode (all switches 0) by the libm and by the sums of setLOD /
                                          * the NCC table, which the PSO trajectory does not depend on. */
} po_scene;

/* mvs/abstractpatch.h:22-53 + patch.h:19-20 */
typedef struct po_patch {
    int      id;
    int      type;
    int      drop;
    int      expanded;
    double   center[3];
    int      numCam;
    int      camIdx[PO_MAX_VIS];
    int      refCamIdx;
    double   normalS[2];
    double   normal[3];
    double   ray[3];
    double   depth;
    double   depthRange[2];
    int      LOD;
    double   imgPoint[PO_MAX_VIS][2];
    double   fitness;
    double   priority;
    double   correlation;
    double   corrTable[PO_MAX_VIS * PO_MAX_VIS];
    /* deterministic-stream bookkeeping (not in the reference) */
    uint64_t key;        /* schedule-independent candidate key */
    int      psoRuns;    /* number of psoOptimization() calls so far  */
    int      psoIters;   /* sum of PsoSolver::getIteration()          */
    int      psoEvals;   /* number of getFitness calls                */
    int      pad0;
    uint64_t psoSig;     /* fold of po_pso_result::gbestSig over the patch's PSO runs */
} po_patch;

/* ---- deterministic uniform stream (replaces rand()/srand(time)) */
uint32_t po_rand31(uint64_t seed, uint64_t key, uint32_t run, uint32_t k);
uint64_t po_child_key(uint64_t parentKey, int cam, int cx, int cy);

/* ---- scene ------------------------------------------------------------- */
void po_config_defaults(po_config *c);              /* TMVS.cpp:26-52            */
void po_config_readme(po_config *c);                /* README.md:110-207         */
void po_camera_init(po_camera *cam, const double focal[2], const double pp[2],
                    const double quat[4], const double center[3]); /* camera.cpp:6-35,108-133 */
int  po_camera_max_lod(int width, int height, double lodRatio, int cfgMaxLOD); /* camera.cpp:63-64 */
po_scene *po_scene_create(const po_config *cfg, int numCams, const po_camera *cams, uint64_t seed);
void po_scene_destroy(po_scene *s);
void po_init_gauss(const po_config *cfg, double *out);   /* mvs.cpp:97-114 */

/* ---- geometry ---------------------------------------------------------- */
void po_spherical2normal(const double s[2], double n[3]);   /* utility.h:25-29 */
void po_normal2spherical(const double n[3], double s[2]);   /* utility.h:17-22 */
int  po_project(const po_scene *s, int cam, const double X[3], double out[2], int LOD); /* camera.cpp:138-160 */
int  po_inv3(const double m[9], double out[9]);             /* OpenCV 2.4 cv::invert 3x3 */
void po_homographies(const po_scene *s, const po_patch *p, const double center[3],
                     const double normal[3], double *H /* numCam*9 */); /* patch.cpp:290-330 */
double po_region_ratio(const po_scene *s, const double pt[2], const double H[9]); /* patch.cpp:269-288 */
void po_fit_ellipse(int n, const float *xy, float *cx, float *cy, float *w, float *h, float *angle); /* OpenCV 2.4 */

double po_exp_det(double x);
double po_exp_poly(double x);   /* exp of the cost weights in kernel-arithmetic mode */ double po_sin_det(double x); double po_cos_det(double x); /* po_detmath.h */

/* ---- cost -------------------------------------------------------------- */
double po_get_fitness(const po_scene *s, const po_patch *p, const double pos[3]); /* patch.cpp:914-1047 */

/* ---- PSO (pso/psosolver.cpp) ------------------------------------------ */
typedef double (*po_fitness_fn)(const double *pos, void *obj);
typedef uint32_t (*po_rand_fn)(void *rngObj);   /* next 31-bit draw */
typedef struct po_pso_result {
    double gBest[3];
    double gBestFitness;
    int    iterations;
    int    evals;
    uint64_t gbestSig;   /* FNV fold of the index of the particle that owns gBest after every updateGbest, and of the
                          * iteration count: two runs with equal signatures took the same discrete trajectory */
} po_pso_result;
/* trace (optional): per iteration, per particle: pos[3] vec[3] pBest[3] fitness pBestFitness, then gIdx, iw */
void po_pso_run(int dim, const double *rangeL, const double *rangeU,
                po_fitness_fn fn, void *obj, int maxIteration, int particleNum,
                const double *init, po_rand_fn rnd, void *rngObj, int omp,
                po_pso_result *res, double *trace, int traceCap, int *traceLen);

/* exported callbacks so the compiled reference PsoSolver (oracle/_ref) can be
 * driven with the restated cost and the deterministic stream */
typedef struct po_fit_ctx { const po_scene *s; const po_patch *p; } po_fit_ctx;
typedef struct po_rng_ctx { uint64_t seed, key; uint32_t run, k; } po_rng_ctx;
double   po_fit_cb(const double *pos, void *obj /* po_fit_ctx* */);
uint32_t po_rng_cb(void *obj /* po_rng_ctx* */);

/* ---- patch state machine (mvs/patch.cpp) ------------------------------ */
void po_patch_init_seed(const po_scene *s, po_patch *p, const double center[3],
                        int numCam, const int *camIdx, uint64_t key);        /* patch.cpp:26-34 */
void po_patch_init_expand(const po_scene *s, po_patch *p, const double center[3],
                          const double parentNormal[3], int parentNumCam,
                          const int *parentCamIdx, uint64_t key);            /* patch.cpp:36-43 */
void po_set_estimated_normal(const po_scene *s, po_patch *p);   /* patch.cpp:390-413 */
void po_set_reference_camera(const po_scene *s, po_patch *p);   /* patch.cpp:415-445 */
void po_set_depth_and_ray(const po_scene *s, po_patch *p);      /* patch.cpp:447-461 */
void po_set_depth_range(const po_scene *s, po_patch *p);        /* patch.cpp:463-509 */
void po_set_lod(const po_scene *s, po_patch *p);                /* patch.cpp:511-610 */
void po_set_priority(const po_scene *s, po_patch *p);           /* patch.cpp:612-625 */
void po_set_image_point(const po_scene *s, po_patch *p);        /* patch.cpp:627-653 */
void po_pso_optimization(const po_scene *s, po_patch *p);       /* patch.cpp:180-219 */
void po_set_correlation_table(const po_scene *s, po_patch *p, const double *H); /* patch.cpp:221-267 */
void po_remove_invisible_camera(const po_scene *s, po_patch *p); /* patch.cpp:655-721 */
void po_expand_visible_camera(const po_scene *s, po_patch *p);   /* patch.cpp:723-761 */
void po_refine(const po_scene *s, po_patch *p);                  /* patch.cpp:114-176 */
int  po_is_neighbor(const po_scene *s, const po_patch *a, const po_patch *b); /* patch.cpp:6-23 */

/* ---- expansion loop (mvs/mvs.cpp) ------------------------------------- */
typedef struct po_mvs po_mvs;
po_mvs *po_mvs_create(po_scene *s);
void    po_mvs_destroy(po_mvs *m);
int     po_mvs_add_seed(po_mvs *m, const double center[3], int numCam, const int *camIdx); /* returns id */
void    po_mvs_set_neighbor_radius(po_mvs *m);          /* mvs.cpp:147-152, 974-997 */
void    po_mvs_refine_seed_patches(po_mvs *m);          /* mvs.cpp:196-231 */
/* Slot-synchronous rounds R(B): an active set of up to B popped
 * parents; every round processes one visible-camera slot of each active parent, in
 * activation order, exactly as the body of MVS::expandNeighborCell does.  B=1 is the
 * reference loop.  maxRounds<=0: run to convergence.  Returns number of refine() calls. */
long    po_mvs_expansion_patches(po_mvs *m, int B, int maxRounds, int strictTail);
/* R(B): a round whose active set has <= thinFront parents handles ALL remaining camera slots of each
 * parent instead of one (0 = never).  Same rule and default as pais_mvs_set_thin_front (include/pais_mvs.h). */
#define PO_DEFAULT_THIN_FRONT 64
void    po_mvs_set_thin_front(po_mvs *m, int thinFront);
/* refine() of the seeds / of the units a round claims is evaluated ahead of the sequential replay, one candidate per OpenMP
 * thread (refine() is a pure function of scene and candidate: same records, same cloud -- tests/test_oracle_parallel.py) */
void    po_mvs_set_parallel(po_mvs *m, int on);
long    po_mvs_speculative(const po_mvs *m);
/* Patch::reCentering (patch.cpp:67-112): imgPoints numCam x 2 pixels -> center */
void    po_recenter(const po_scene *s, int numCam, const int *camIdx, const double *imgPoints, double center[3]);
/* post filters (`-f` verb, TMVS.cpp:124-172; mvs.cpp:278-524) and the .mvs loader constructor (patch.cpp:45-59) */
int     po_mvs_load_patch(po_mvs *m, const double center[3], const double normalS[2], int numCam, const int *camIdx,
                          double fitness, double correlation);
void    po_mvs_cell_filtering(po_mvs *m);
void    po_mvs_visibility_filtering(po_mvs *m);
void    po_mvs_neighbor_cell_filtering(po_mvs *m, double neighborRatio);
void    po_mvs_neighbor_patch_filtering(po_mvs *m, double neighborRatio, int *counts);
int     po_mvs_num_patches(const po_mvs *m);
int     po_mvs_num_slots(const po_mvs *m);              /* ids are 0..slots-1 */
const po_patch *po_mvs_get_patch(const po_mvs *m, int id); /* NULL if deleted */
long    po_mvs_refine_calls(const po_mvs *m);
long    po_mvs_fitness_evals(const po_mvs *m);
int     po_runtime_filtering(const po_mvs *m, const po_patch *p);   /* mvs.cpp:838-898 */
void    po_expansion_center(const po_scene *s, int cam, const po_patch *parent, int cx, int cy, double out[3]); /* mvs.cpp:809-836 */
/* single-candidate helper for per-candidate parity: constructs + refines +
 * removeInvisibleCamera exactly as MVS::expandCell (mvs.cpp:566-577) */
void    po_expand_candidate(const po_scene *s, po_patch *out, const double center[3],
                            const double parentNormal[3], int parentNumCam,
                            const int *parentCamIdx, uint64_t key);
void    po_expand_candidates_parallel(const po_scene *s, po_patch *out, int n, const double *centers,
                                      const double *parentNormals, const int *parentNumCam, const int *parentCamIdx,
                                      const uint64_t *keys); /* parentCamIdx: PO_MAX_VIS ints per candidate */
void    po_refine_seed(const po_scene *s, po_patch *p);  /* refine + removeInvisibleCamera, mvs.cpp:214-215 */

size_t  po_sizeof_patch(void);
size_t  po_sizeof_config(void);
size_t  po_sizeof_camera(void);

#ifdef __cplusplus
}
#endif
#endif
