/*
 * pais_mvs.h -- C ABI of the host-side reconstruction driver: the MI355X
 * drop-in behind MVS::refineSeedPatches() / MVS::expansionPatches()
 * (TMVS/mvs/mvs.h:229,231; TMVS/mvs/mvs.cpp:196-275, 529-898).
 *
 * The driver owns what the reference's MVS singleton owns on this path --
 * patches (map<int,Patch>, mvs.h:86), one CellMap per camera (mvs.h:88,
 * cellmap.h:15-32), the priority queue (mvs.h:92), neighborRadius -- and
 * forwards every batch of constructed-but-unrefined patches to
 * pais_refine_batch() (include/pais_hip.h).
 *
 * Expansion runs in slot-synchronous rounds R(B): an ordered
 * active set of up to B parents popped with the reference's queue policy; every
 * round handles ONE visible-camera slot of each active parent -- all candidate
 * cells of the round are refined speculatively in one GPU batch, then the
 * reference's sequential steps (skipNeighborCell, expandCell, insertPatch:
 * mvs.cpp:552-562, 566-601) are replayed on the host in activation order,
 * consuming only the candidates the sequential order evaluates.  The accepted
 * cloud is identical to running that schedule one candidate at a time (the
 * oracle does exactly that), independent of how a batch was split across GPUs,
 * and B = 1 is the reference's own order.
 *
 * The stepwise entry points (round_begin / round_commit) exist so that several
 * ranks, each with its own GPU and a replicated driver, can refine disjoint
 * shards of a round's candidates and exchange the records with one all-gather.
 */
#ifndef PAIS_MVS_H
#define PAIS_MVS_H

#include "pais_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pais_mvs pais_mvs;

typedef struct pais_mvs_stats {
    int64_t seeds_refined;        /* refine() calls on seeds                                   */
    int64_t candidates_refined;   /* expansion candidates sent to the GPU (speculative superset) */
    int64_t candidates_effective; /* of those, the ones the sequential order evaluates          */
    int64_t patches_inserted;     /* insertPatch() successes                                    */
    int64_t patches_deleted;
    int64_t rounds;
    int64_t parents_popped;
    int64_t pso_evals_effective;  /* getFitness calls of the effective refines                  */
    double  host_enumerate_ms, host_commit_ms, gpu_refine_ms;
    int64_t batches_sharded;      /* batches whose candidates were split across the ranks (one all-gather each) */
    int64_t batches_replicated;   /* multi-rank batches too small to split: every rank refined all of them      */
    double  exchange_ms;          /* host time spent in the all-gathers (incl. waiting for the slowest rank)    */
    int64_t exchange_bytes;       /* bytes a rank received in them: world x (64-byte header + shard x wire slot) per batch */
    int64_t rounds_streamed;      /* rounds whose work list went to the GPU in two parts (the second enumerated, the first
                                   * committed, while the other was being refined); host_enumerate_ms / host_commit_ms then
                                   * include host work that ran under the GPU's, gpu_refine_ms only what the host waited for */
    double  emu_replay_ms;        /* pais_mvs_emulate mode 2: host time spent packing the OTHER ranks' blocks from the recorded
                                   * run -- work a real rank does not have; bench.py subtracts it from the emulated rank's time */
    int64_t exchange_retries;     /* sharded batches that took a SECOND exchange: some rank's one-launch PSO pass (k_pso_ring) did not
                                   * complete and its shard was refined again.  Correct, but never expected: bench.py --gpus N > 1
                                   * refuses to print a line when it happened */
    int64_t rounds_enum_sharded;  /* rounds whose listing (skip test + claim of the units) was dealt to the ranks by (camera, tile) and
                                   * merged with one all-gather of the unit states (PAIS_SHARD_ENUM=1; off by default) */
} pais_mvs_stats;

/* One entry per GPU batch of the last reconstruction (the seed batch first, then one per expansion round with
 * candidates): what a scaling model needs -- how many candidates, whether the batch was sharded, and the host / GPU
 * times around it (milliseconds).  bench.py prints a predicted speed-up at 2 / 4 / 8 GPUs from it. */
typedef struct pais_round_log {
    int32_t n;                    /* candidates of the batch                                     */
    int32_t has_seeds;
    int32_t sharded;              /* 1: split across the ranks, 0: one rank / replicated         */
    int32_t max_num_cam;
    double  refine_ms;            /* pais_refine_batch (or shard + exchange) as the host saw it  */
    double  enumerate_ms, commit_ms; /* host work before / after it (0 for the seed batch's enumerate) */
} pais_round_log;
/* copies up to `cap` entries into out (may be NULL); returns the number of entries recorded */
int  pais_mvs_get_round_log(const pais_mvs *m, pais_round_log *out, int cap);

/* Creates the driver and its own pais_ctx on `device` (MVS::getInstance(config),
 * mvs.cpp:22-34 + loading the cameras).  device < 0 creates the scheduler without
 * a GPU context: only the stepwise entry points work then (records must come from
 * a rank that has a GPU); nothing is ever computed on the host instead. */
int  pais_mvs_create(const pais_config *cfg, int num_cams, const pais_camera_desc *cams,
                     int device, uint64_t pso_seed, pais_mvs **out);
void pais_mvs_destroy(pais_mvs *m);
pais_ctx *pais_mvs_ctx(pais_mvs *m);
/* forget all patches, cell maps and the queue; cameras and the GPU context stay */
int  pais_mvs_reset(pais_mvs *m);

/* Seed constructor Patch(center, color, camIdx, imgPoint) (patch.cpp:26-34,
 * setEstimatedNormal :390-413).  Returns the patch id (>= 0) or < 0. */
int  pais_mvs_add_seed(pais_mvs *m, const double center[3], int num_cam, const int32_t *cam_idx);

/* NVM seed with its measurements (image points in pixels, num_cam x 2: FileLoader::loadNvmPatch adds cols/2, rows/2
 * to the file's offsets, fileloader.cpp:155-160).  recenter != 0: MVS::reCentering first (mvs.cpp:134-145,
 * patch.cpp:67-112) -- the point closest to the viewing rays, A.inv(DECOMP_SVD) * b -- as loadNVM does. */
int  pais_mvs_add_seed_measured(pais_mvs *m, const double center[3], int num_cam, const int32_t *cam_idx,
                                const double *img_points, int recenter);

/* MVS::refineSeedPatches (mvs.cpp:196-231), one GPU batch. */
int  pais_mvs_refine_seed_patches(pais_mvs *m);
/* MVS::expansionPatches (mvs.cpp:233-275) in rounds of `parents_per_round`
 * parents; max_rounds <= 0: until the queue is empty. */
int  pais_mvs_expansion_patches(pais_mvs *m, int parents_per_round, int max_rounds);

/* Round rule for the tail of the expansion: a round whose active set holds <= thin_front parents handles
 * ALL remaining camera slots of each of them instead of one (0 = never).  Thin rounds are pure latency on
 * the GPU, so taking the whole parent costs a few speculative refines and saves 4-5 rounds per generation.
 * With one parent per round this is exactly MVS::expandNeighborCell's loop over the visible cameras
 * (mvs.cpp:529-563).  Part of the schedule definition R(B): the oracle applies the same rule. */
#define PAIS_DEFAULT_THIN_FRONT 64
int  pais_mvs_set_thin_front(pais_mvs *m, int thin_front);

/* ---- multi-GPU: one process per GPU, replicated driver, sharded refinement ----
 * Every rank creates the same driver on its own GPU (same cameras, config, seeds, pso_seed) and joins a
 * communicator.  From then on pais_mvs_refine_seed_patches / pais_mvs_expansion_patches split every batch of
 * candidates into `world` contiguous, count-balanced shards (PAIS_SHARD_STRIDED=1: candidate i to rank i mod world instead); a rank refines its shard (pais_refine_batch_device_async,
 * records stay in HBM) and the records -- packed into fixed-size wire slots behind a 64-byte status header that is written on
 * the device -- are exchanged with ONE all-gather per batch -- ncclAllGather (RCCL over xGMI) on the context's stream, ONE host
 * synchronisation per batch -- after which every rank replays the identical host bookkeeping.  Large rounds are streamed in
 * PAIS_STREAM_PARTS (4) sharded parts on lanes, so that the replicated host work runs under the GPUs' work; whether a round
 * is streamed is rank 0's timing-driven choice, carried in the headers, so that every rank issues the same collectives.  Batches of fewer than PAIS_REPLICATE_BELOW_WAVES evaluation waves per PSO iteration (candidates x
 * particles) are latency bound on one GPU already -- their launches take one wave's latency however many GPUs share
 * them -- so every rank refines all of them itself (the refinement is deterministic: the replicas agree bit for bit) and
 * no collective is issued.  The cloud is identical for every world size.
 * This is what shards MVS::expansionPatches (mvs.cpp:233-275) and MVS::refineSeedPatches (:196-231). */
#define PAIS_UNIQUE_ID_BYTES 128          /* == NCCL_UNIQUE_ID_BYTES */
#define PAIS_REPLICATE_BELOW_WAVES 1024   /* 68 expansion candidates / 34 seeds at particleNum 15 */
typedef struct pais_unique_id { char bytes[PAIS_UNIQUE_ID_BYTES]; } pais_unique_id;
/* ncclGetUniqueId: call on ONE rank, hand the 128 bytes to the others (MPI_Bcast, a file, a TCP store ...) */
int  pais_comm_get_unique_id(pais_unique_id *out);
/* ncclCommInitRank on the driver's GPU (collective over all ranks of the job). */
int  pais_mvs_comm_init_rccl(pais_mvs *m, int rank, int world, const pais_unique_id *id);
/* pais_mvs_create + pais_mvs_comm_init_rccl in one call (device = this rank's GPU). */
int  pais_mvs_create_ranked(const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device,
                            uint64_t pso_seed, int rank, int world, const pais_unique_id *id, pais_mvs **out);
/* Any other transport (MPI, a test's host-memory all-gather for several ranks that share ONE GPU, which RCCL
 * refuses): send = this rank's bytes_per_rank bytes, recv = world * bytes_per_rank bytes in rank order, both HOST
 * pointers; returns 0 on success.  The driver stages the records through pinned host memory around the call. */
typedef int (*pais_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int  pais_mvs_comm_init_callback(pais_mvs *m, int rank, int world, pais_allgather_fn fn, void *user);
/* Evaluation waves per iteration below which a multi-rank batch is replicated instead of sharded (default
 * PAIS_REPLICATE_BELOW_WAVES; 0 = always shard).  Must be the same on every rank. */
int  pais_mvs_set_replicate_below(pais_mvs *m, int waves);
/* (test / measurement hooks -- pais_mvs_emulate, pais_mvs_set_record_source -- live in include/pais_test_hooks.h: a reference
 * maintainer binding this library needs neither) */

/* ---- stepwise ---------------------------------------------------------- */
/* seeds: candidates of all seeds with camNum >= minCamNum (others are deleted) */
int  pais_mvs_seed_begin(pais_mvs *m, const pais_candidate **cands, int *n);
int  pais_mvs_seed_commit(pais_mvs *m, const pais_patch_result *results, int n);
/* setCellMaps + initPriorityQueue + setNeighborRadius (mvs.cpp:235-239) */
int  pais_mvs_expansion_begin(pais_mvs *m);
/* returns 0 with *n >= 0 candidates (pointer valid until the next call), 1 when expansion is finished */
int  pais_mvs_round_begin(pais_mvs *m, int parents_per_round, const pais_candidate **cands, int *n);
int  pais_mvs_round_commit(pais_mvs *m, const pais_patch_result *results, int n);
int  pais_mvs_expansion_end(pais_mvs *m);   /* setNeighborRadius (mvs.cpp:274) */

/* ---- post filters: the `-f` verb (TMVS.cpp:124-172) --------------------- */
/* Loader constructor Patch(center, normalS, camIdx, fitness, correlation, id) (patch.cpp:45-59), i.e. what
 * FileLoader::loadMvsPatch builds from a .mvs file (fileloader.cpp:206-231).  Returns the id or < 0. */
int  pais_mvs_load_patch(pais_mvs *m, const double center[3], const double normalS[2], int num_cam,
                         const int32_t *cam_idx, double fitness, double correlation);
/* MVS::cellFiltering / visibilityFiltering / neighborCellFiltering (mvs.cpp:278-446): sequential host passes in
 * the reference's order (their decisions depend on the deletions made so far). */
int  pais_mvs_cell_filtering(pais_mvs *m);
int  pais_mvs_visibility_filtering(pais_mvs *m);
int  pais_mvs_neighbor_cell_filtering(pais_mvs *m, double neighbor_ratio);
/* MVS::neighborPatchFiltering (mvs.cpp:448-524): all-pairs neighbour counts on the GPU (k_neighbor_count), then
 * the reference's average / ratio rule.  kernel_ms (optional) returns the kernel's duration. */
int  pais_mvs_neighbor_patch_filtering(pais_mvs *m, double neighbor_ratio, double *kernel_ms);

/* ---- inspection -------------------------------------------------------- */
int    pais_mvs_num_patches(const pais_mvs *m);
int    pais_mvs_num_slots(const pais_mvs *m);        /* ids are 0 .. slots-1 */
/* returns 0 and fills *out (+ *expanded) if the id is alive, 1 if deleted */
int    pais_mvs_get_patch(const pais_mvs *m, int id, pais_patch_result *out, int *expanded);
double pais_mvs_neighbor_radius(const pais_mvs *m);
int    pais_mvs_get_stats(const pais_mvs *m, pais_mvs_stats *out);
const char *pais_mvs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
