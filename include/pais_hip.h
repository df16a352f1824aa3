/*
 * pais_hip.h -- C ABI of the MI355X-native PAIS-MVS refine/expansion hot path.
 *
 * This is the drop-in boundary.
 * The reference has no FFI layer; the seams this library replaces are
 *
 *   - the C-style cost callback  double (*getFitness)(const Particle&, void*)
 *     bound at  TMVS/mvs/patch.cpp:192,199  (declared pso/psosolver.h:72)
 *       -> pais_fitness_batch()
 *   - Patch::refine() + Patch::removeInvisibleCamera() as called from
 *     MVS::refineSeedPatches (TMVS/mvs/mvs.cpp:214-215) and MVS::expandCell
 *     (TMVS/mvs/mvs.cpp:573-574)
 *       -> pais_refine_batch()
 *   - the scene state those read through the MVS singleton
 *     (TMVS/mvs/mvs.h:84-94,196-223; TMVS/mvs/camera.h:79-106)
 *       -> pais_ctx_create() / pais_ctx_set_config() / pais_ctx_set_neighbor_radius()
 *
 * Conventions: plain C, POD structs, double precision, caller owns every
 * buffer, the library keeps no pointer past a call (scene data is copied to
 * the GPU in pais_ctx_create).  Functions return 0 on success, <0 on argument
 * / HIP failure (text via pais_last_error()).  Numerical invalidity stays
 * in-band exactly as in the reference: DBL_MAX fitness (patch.cpp:940,953,
 * 961,1001), NaN passthrough, `dropped` flag (patch.cpp:119-122).
 *
 * INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef PAIS_HIP_H
#define PAIS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAIS_MAX_LEVELS 16  /* LOD 0..15; MvsConfig::maxLOD <= 15 (TMVS.cpp:42) */
#define PAIS_MAX_VIS    64  /* visible cameras tracked per patch (camIdx).  The reference's vector<int> is unbounded;
                             * a patch whose visibility cone (patch.cpp:723-761) holds more cameras than this is an
                             * error here (pais_mvs_round_begin fails), never a silent truncation. */
#define PAIS_MAX_PARTICLES 128 /* 2*particleNum for seeds must fit             */

#define PAIS_TYPE_SEED   0  /* Patch::TYPE_SEED,   mvs/patch.h:17 */
#define PAIS_TYPE_EXPAND 1  /* Patch::TYPE_EXPAND, mvs/patch.h:18 */

/* PAIS::MvsConfig, TMVS/mvs/mvs.h:19-72 (same fields, same meaning; the three
 * bools are widened to int32 so the struct has no padding surprises). */
typedef struct pais_config {
    int32_t cellSize;
    int32_t patchRadius;
    int32_t patchSize;              /* ignored on input: always 2*patchRadius+1 (mvs.cpp:67) */
    int32_t minCamNum;
    double  textureVariation;
    double  visibleCorrelation;
    double  minCorrelation;
    double  maxFitness;
    double  lodRatio;
    int32_t minLOD;
    int32_t maxLOD;
    int32_t maxCellPatchNum;
    int32_t _pad0;
    double  reduceNormalRange;
    int32_t adaptiveDistanceEnable;
    int32_t adaptiveDifferenceEnable;
    int32_t adaptiveGradientEnable;
    int32_t _pad1;
    double  distWeighting;
    double  diffWeighting;
    double  gradientWeighting;
    double  neighborRadius;
    double  neighborRadiusScalar;
    double  minRegionRatio;
    double  depthRangeScalar;
    int32_t particleNum;
    int32_t maxIteration;
    int32_t expansionStrategy;
    int32_t _pad2;
} pais_config;

/* What the hot path reads from PAIS::Camera (TMVS/mvs/camera.h:79-106).  The
 * caller passes the reference's own matrices and pyramid buffers; nothing is
 * recomputed here. */
typedef struct pais_camera_desc {
    double focal[2];             /* getFocalLength()                        */
    double principle_point[2];   /* getPrinciplePoint()                     */
    double rotation[9];          /* getRotation(), row-major 3x3            */
    double translation[3];       /* getTranslation()                        */
    double center[3];            /* getCenter()                             */
    double KR[9];                /* getKR()                                 */
    double KT[3];                /* getKT()                                 */
    double optical_normal[3];    /* getOpticalNormal()                      */
    int32_t max_lod;             /* getMaxLOD()                             */
    int32_t _pad;
    int32_t level_width[PAIS_MAX_LEVELS];    /* getPyramidImage(l).cols     */
    int32_t level_height[PAIS_MAX_LEVELS];   /* getPyramidImage(l).rows     */
    int64_t level_stride[PAIS_MAX_LEVELS];   /* Mat::step in bytes (0 = width) */
    const uint8_t *level_image[PAIS_MAX_LEVELS]; /* getPyramidImage(l).data, host memory */
    const double  *level_edge[PAIS_MAX_LEVELS];  /* getPyramidEdge(l).data (dense rows), host memory.  May be NULL
                                                    (all levels of all cameras): with adaptiveGradientEnable set the
                                                    library then evaluates the edge maps on the fly from the gray
                                                    levels with the statements of camera.cpp:72-77,87-91 (Sobel ksize 1,
                                                    magnitude, per-level min-max) -- no 8-byte-per-pixel copy in HBM */
} pais_camera_desc;

/* The part of a Patch that PAIS::getFitness reads (patch.cpp:922-944). */
typedef struct pais_patch_state {
    double  ray[3];              /* getRay()                    */
    int32_t ref_cam;             /* getReferenceCameraIndex()   */
    int32_t lod;                 /* getLOD()                    */
    int32_t num_cam;             /* getCameraNumber()           */
    int32_t _pad;
    int32_t cam_idx[PAIS_MAX_VIS]; /* getCameraIndices()        */
} pais_patch_state;

/* A constructed-but-unrefined patch: the state right after the seed
 * constructor (patch.cpp:26-34) or the expansion constructor (patch.cpp:36-43,
 * i.e. parent normal inherited and expandVisibleCamera() applied). */
typedef struct pais_candidate {
    double   center[3];
    double   normal[3];
    double   normalS[2];         /* spherical form of `normal` (abstractpatch.cpp:43-46) */
    uint64_t key;                /* schedule-independent key of the deterministic PSO stream */
    int32_t  type;               /* PAIS_TYPE_SEED / PAIS_TYPE_EXPAND */
    int32_t  num_cam;
    int32_t  cam_idx[PAIS_MAX_VIS];
} pais_candidate;

/* The patch after refine() + removeInvisibleCamera() (AbstractPatch fields,
 * mvs/abstractpatch.h:22-53).  When `dropped` != 0 only `dropped` is meaningful. */
typedef struct pais_patch_result {
    double   center[3];
    double   normal[3];
    double   normalS[2];
    double   ray[3];
    double   depth;
    double   depthRange[2];
    double   fitness;
    double   priority;
    double   correlation;
    double   imgPoint[PAIS_MAX_VIS][2];
    uint64_t key;
    int32_t  type;
    int32_t  dropped;
    int32_t  num_cam;
    int32_t  ref_cam;
    int32_t  lod;
    int32_t  pso_runs;           /* psoOptimization() calls                 */
    int32_t  pso_iterations;     /* sum of PsoSolver::getIteration()        */
    int32_t  pso_evals;          /* getFitness evaluations executed         */
    int32_t  cam_idx[PAIS_MAX_VIS];
    /* refine() loop bookkeeping (patch.cpp:132-171); internal, kept in the
     * record so that a record is the complete device-side state */
    int32_t  stage;
    int32_t  before_ref, before_num, after_ref, after_num, count, total_cam_num;
    int32_t  ncc_tables;         /* setCorrelationTable() calls             */
} pais_patch_result;

/* pais_patch_result.stage of a returned record (other values are internal to a batch call).
 * The scene half of MVS::runtimeFiltering (mvs.cpp:851-863: EVERY camera of the scene projects
 * the centre inside its image, on a non-background pixel) depends on the record and the scene
 * alone; the batch evaluates it on the device -- one lane per camera -- so that the host test
 * that follows every refine (mvs.cpp:583) keeps only its cell-map half. */
#define PAIS_DONE            0  /* finished; scene test not evaluated (dropped records, host-built records) */
#define PAIS_DONE_IN_SCENE   5  /* finished, not dropped, the scene test passes */
#define PAIS_DONE_OFF_SCENE  6  /* finished, not dropped, the scene test fails  */

typedef struct pais_ctx pais_ctx;

/* Create the per-scene context on HIP device `device`: copies config, camera
 * matrices, gray pyramids (and edge pyramids if given) into HBM and builds the
 * Gaussian distance table (MVS::initPatchDistanceWeighting, mvs.cpp:97-114).
 * `pso_seed` seeds the deterministic uniform stream that replaces
 * srand(time)+rand() (psosolver.cpp:60-68). */
int  pais_ctx_create(const pais_config *cfg, int num_cams, const pais_camera_desc *cams,
                     int device, uint64_t pso_seed, pais_ctx **out);
/* MVS::setConfig (mvs.cpp:42-72): replaces the config, rebuilds the table. */
int  pais_ctx_set_config(pais_ctx *ctx, const pais_config *cfg);
/* MVS::setNeighborRadius result (mvs.cpp:147-152); read by setDepthRange (patch.cpp:508). */
int  pais_ctx_set_neighbor_radius(pais_ctx *ctx, double neighbor_radius);
void pais_ctx_destroy(pais_ctx *ctx);

/* out[e] = PAIS::getFitness(particle e, &patch[state_index[e]])  (patch.cpp:914-1047).
 * particles: n_evals x 3 doubles (theta, phi, depth).  Host pointers. */
int  pais_fitness_batch(pais_ctx *ctx, int n_states, const pais_patch_state *states,
                        int n_evals, const int32_t *state_index, const double *particles,
                        double *out);

/* For each candidate: Patch::refine() followed by Patch::removeInvisibleCamera()
 * (mvs.cpp:214-215 / 573-574).  Host pointers; synchronous at return. */
int  pais_refine_batch(pais_ctx *ctx, int n, const pais_candidate *cands, pais_patch_result *out);
/* Same without the last host copy: *view points at the n records in the context's
 * pinned staging buffer, valid until the next batch call on this context (what
 * MVS::refineSeedPatches / expansionPatches need: they read every record once). */
int  pais_refine_batch_view(pais_ctx *ctx, int n, const pais_candidate *cands, const pais_patch_result **view);
/* pais_refine_batch_view in two halves.  _begin copies the candidates to pinned memory and enqueues the whole batch
 * (copy up, refine, copy down); for a pure expansion batch it returns as soon as that is enqueued (a batch that holds
 * seeds is refined before it returns: the seed loop takes host decisions).  _end waits for the batch and hands out the
 * view.  Exactly one batch can be open per context; the caller's `cands` are not read after _begin returns. */
int  pais_refine_batch_begin(pais_ctx *ctx, int n, const pais_candidate *cands);
int  pais_refine_batch_end(pais_ctx *ctx, const pais_patch_result **view);
/* ---- ADVANCED: launch-chain scheduling (pais_refine_batch_open / _enqueue, pais_ctx_fork_lane, pais_ctx_set_round_hint).
 * A maintainer who binds MVS::refineSeedPatches / expansionPatches never needs these four: pais_refine_batch(_view) and
 * the drivers of include/pais_mvs.h are the whole drop-in surface.  They exist for a driver that overlaps its own host
 * work with the GPU's (the streamed rounds of pais_mvs_expansion_patches are built on them) and are exported so that
 * such a driver can live outside this library. ----
 * _begin cut in pieces, for a driver that keeps two lanes busy: the launches of a batch are a chain per PSO iteration,
 * and a chain that is enqueued in one go keeps the host away from the other lane for its whole length.  _open = _begin
 * that stops after the first `iterations` PSO iterations; _enqueue adds up to `iterations` more (<= 0: all that remain)
 * and returns 1 once the batch is enqueued to its end (after-stage and copy down included), 0 while iterations remain,
 * < 0 on error.  _end enqueues whatever remains. */
int  pais_refine_batch_open(pais_ctx *ctx, int n, const pais_candidate *cands, int iterations);
int  pais_refine_batch_enqueue(pais_ctx *ctx, int iterations);
/* A second lane over the same scene: a context that shares the parent's cameras, pyramids and configuration (nothing
 * is copied or re-uploaded) and has its own stream, work buffers and staging, so that a batch can be open on each --
 * the drivers of include/pais_mvs.h enumerate the second half of a round while the first half is being refined.
 * The lane belongs to the parent: pais_ctx_destroy(parent) releases it, pais_ctx_set_config / _set_neighbor_radius /
 * _set_fine_timing on the parent apply to it, pais_get_kernel_stats(parent) includes its launches.  Records are the
 * same bit for bit whichever lane refines a candidate. */
int  pais_ctx_fork_lane(pais_ctx *parent, pais_ctx **lane);
/* The next batch of this context is one of several parts of a round of about `n_round` candidates that are refined at
 * the same time (on other lanes): the launch shapes -- which PSO pipeline, how many sub-streams, how many waves per
 * evaluation -- are chosen for the round, not for the part.  Applies to one batch; never changes a record. */
int  pais_ctx_set_round_hint(pais_ctx *ctx, int n_round);

/* Same, with device pointers (results stay in HBM, e.g. as the send buffer of
 * the per-round RCCL all-gather).  Work is enqueued on the context's stream
 * (pais_ctx_stream()).  `max_num_cam` bounds num_cam over the batch (<=0: use
 * min(num_cams, PAIS_MAX_VIS)); `has_seeds` != 0 if any candidate is a seed
 * (the seed loop needs one host decision per pass and is then synchronous;
 * pure expansion batches are enqueued asynchronously). */
int  pais_refine_batch_device(pais_ctx *ctx, int n, const pais_candidate *d_cands,
                              pais_patch_result *d_out, int max_num_cam, int has_seeds);
/* ---- wire format of a record (the per-round exchange of the multi-GPU drivers, include/pais_mvs.h) ----
 * A pais_patch_result is 1488 bytes, 1280 of them the two PAIS_MAX_VIS-long arrays.  What a driver's commit reads of a
 * record of a batch whose candidates see at most `max_num_cam` cameras fits in
 *     pais_record_wire_bytes(max_num_cam) = 208 + 20 * roundup2(max_num_cam)       (308 bytes at 5 cameras)
 * [17 doubles center .. correlation][key][type .. pso_evals][stage .. ncc_tables][imgPoint[Kw][2]][cam_idx[Kw]],
 * every slot of a batch the same size (an all-gather needs that).  Unpacking zero-fills the array tails beyond Kw -- the
 * batch calls never write an array element at or beyond the candidate's own camera count, so for records of such a
 * batch pack -> unpack reproduces every byte.
 * pack / unpack are plain host loops; _device is the same packing by a kernel on the context's stream (d_* device
 * pointers). */
size_t pais_record_wire_bytes(int max_num_cam);
int  pais_pack_records(int n, const pais_patch_result *recs, int max_num_cam, void *wire);
int  pais_unpack_records(int n, const void *wire, int max_num_cam, pais_patch_result *recs);
int  pais_pack_records_device(pais_ctx *ctx, int n, const pais_patch_result *d_recs, int max_num_cam, void *d_wire);

/* ---- ADVANCED: what a multi-GPU driver needs to keep ONE synchronisation per sharded batch (the drivers of
 * include/pais_mvs.h use them; a maintainer who binds those drivers never calls them) ----
 * pais_refine_batch_device_async: pais_refine_batch_device that never waits for a pure expansion batch, also when its PSO
 * pass runs as k_pso_ring (whose completion words pais_refine_batch_device reads before it returns).
 * pais_wire_header_device: the 64-byte status header of this rank's block of the exchange, written on the device behind the
 * batch's launches: uint32 {PAIS_WIRE_MAGIC, rc, count, rank, user_word, 0 x 11}; rc = host_rc if non-zero,
 * PAIS_WIRE_RC_RING_RETRY if the batch's ring pass did not complete, else 0; user_word travels as given (the drivers of
 * include/pais_mvs.h carry rank 0's timing-driven scheduling choice for the next round in it, so that every rank takes the
 * same one).
 * pais_ctx_batch_status: after the caller synchronised with the stream: 0 = the last _async batch is complete; 1 = its ring
 * pass did not complete: the records are invalid and the caller refines the same batch again (that batch then takes one
 * launch per iteration).  Every rank reads every header, so all ranks agree on a second exchange. */
#define PAIS_WIRE_MAGIC 0x50414953u
#define PAIS_WIRE_RC_RING_RETRY 1
#define PAIS_WIRE_HEADER_BYTES 64
int  pais_refine_batch_device_async(pais_ctx *ctx, int n, const pais_candidate *d_cands,
                                    pais_patch_result *d_out, int max_num_cam, int has_seeds);
int  pais_wire_header_device(pais_ctx *ctx, int rank, int count, int host_rc, uint32_t user_word, void *d_header);
int  pais_ctx_batch_status(pais_ctx *ctx);

/* The HIP stream (hipStream_t) all work of this context is enqueued on. */
void *pais_ctx_stream(pais_ctx *ctx);
int  pais_ctx_synchronize(pais_ctx *ctx);

/* Kernel timing accumulated since the last reset, measured with HIP events on the
 * launch stream.  pso_ms spans one whole PSO pass (k_pso_init + the pass's evaluation /
 * step launches: k_pso_iter per iteration for small batches, k_pso_ring -- one launch --
 * or k_pso_eval2 + k_pso_step per iteration for large ones, k_pso_tile for many cameras);
 * eval_ms / eval_launches cover every cost-evaluation launch and are only collected while
 * fine timing is on (pais_ctx_set_fine_timing, or PAIS_FINE_TIMING=1 in the environment;
 * bench.py's roofline leg).  Algorithmic bytes:
 * SURVEY 8d, S^2*(4K+1+8[dist]+8[grad]) per cost evaluation. */
typedef struct pais_kernel_stats {
    double   pso_ms;
    double   begin_ms;
    double   after_ms;
    int64_t  pso_launches;
    int64_t  pso_evals;
    int64_t  pso_patches;
    double   pso_algorithmic_bytes;
    double   ncc_algorithmic_bytes;
    int64_t  ncc_tables;
    double   eval_ms;
    int64_t  eval_launches;
    /* the large-batch evaluation kernel k_pso_eval2 ALONE (a subset of eval_ms / eval_launches / pso_evals /
     * pso_algorithmic_bytes): its launches are throughput bound, the k_pso_iter launches of small batches latency bound */
    double   eval2_ms;
    int64_t  eval2_launches;
    int64_t  eval2_evals;
    double   eval2_algorithmic_bytes;
    int64_t  tile_launches;      /* of eval2_launches: evaluations by the LDS-tile kernel k_pso_tile (many-camera batches) */
    /* the launches of eval2_ms run on two (streamed rounds: up to four) streams at once, which stretches each of them:
     * eval2_busy_ms is the length of the UNION of their [start, end] intervals -- the time during which at least one of
     * them was running (<= eval2_ms; 0 unless fine timing was on) */
    double   eval2_busy_ms;
    int64_t  ring_launches;      /* of eval2_launches: whole PSO passes run by k_pso_ring (one launch each) */
    /* the k_pso_ring launches ALONE (a subset of the eval2_* figures): sum of their durations (fine timing), the cost
     * evaluations they ran and their algorithmic bytes -- what bench.py's roofline object is computed from */
    double   ring_ms;
    int64_t  ring_evals;
    double   ring_algorithmic_bytes;
    int64_t  ring_fallbacks;     /* ring passes that did not complete (a wave waited longer than PAIS_RING_TIMEOUT_MS) and
                                  * were re-run through the per-iteration launches: same records, 0 in every healthy run */
} pais_kernel_stats;
int  pais_get_kernel_stats(pais_ctx *ctx, pais_kernel_stats *out, int reset);
/* on != 0: bracket every cost-evaluation launch (k_pso_iter / k_fitness) with HIP events on the stream it is
 * launched on; their durations accumulate in eval_ms / eval_launches.  Off by default (two event records per
 * launch are not free); the environment variable PAIS_FINE_TIMING=1 turns it on at context creation. */
int  pais_ctx_set_fine_timing(pais_ctx *ctx, int on);

/* MVS::neighborPatchFiltering's O(n^2) part (mvs.cpp:448-524): counts[i] = number of OTHER centres within `radius`
 * of centre i (dist = sqrt(dx^2+dy^2+dz^2) in that order, counted when !(dist > radius)).  centers: n x 3 doubles on
 * the host.  kernel_ms (optional): duration of the kernel alone. */
int  pais_neighbor_count(pais_ctx *ctx, int n, const double *centers, double radius, int32_t *counts, double *kernel_ms);

/* Deterministic stream helpers (shared by host scheduler and tests). */
uint32_t pais_rand31(uint64_t seed, uint64_t key, uint32_t run, uint32_t k);
uint64_t pais_child_key(uint64_t parent_key, int cam, int cx, int cy);

const char *pais_last_error(void);
size_t pais_sizeof_config(void);
size_t pais_sizeof_camera_desc(void);
size_t pais_sizeof_candidate(void);
size_t pais_sizeof_patch_result(void);

#ifdef __cplusplus
}
#endif
#endif /* PAIS_HIP_H */
