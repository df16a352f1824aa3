// pais_capi.hip -- the C ABI of include/pais_hip.h: scene upload, batch entry
// points, kernel timing.  No algorithmic fallbacks live here: every entry point
// either runs the gfx950 kernels or returns an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>
#include <algorithm>

#include "../../include/pais_hip.h"
#include "pais_dev.hpp"
#include "pais_internal.h"

static thread_local std::string g_err;
static int fail(const char *what, hipError_t e)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    g_err = buf;
    return -2;
}
static int fail_msg(const char *what)
{
    g_err = what;
    return -1;
}
#define HIPCHK(call)                                   \
    do {                                               \
        hipError_t e__ = (call);                       \
        if (e__ != hipSuccess) return fail(#call, e__); \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
};

struct pais_ctx;
// brackets a region of a stream with an event pair while profiling is on (defined below)
struct Timed {
    pais_ctx *ctx = nullptr;
    hipStream_t st = nullptr;
    std::vector<EventPair> *into = nullptr;
    EventPair e{nullptr, nullptr};
    bool on = false;
    int begin(pais_ctx *c, hipStream_t s, std::vector<EventPair> *v);
    int end();
};
// a PSO pass of a batch between pass_open and pass_close (refine batch section)
struct PassPlan {
    struct Slice { int lo, hi, parts; hipStream_t st; bool own; };
    int n = 0, Nmax = 0, Kmax = 0, maxIt = 0, nSl = 0, S = 1, afterGrid = 0, itNext = 0;
    bool useIter = false, useTile = false, useRing = false, hasSeeds = false, hostBatch = false;
    bool usePre = false; // the swarm step writes the evaluations' set-up records (pais_pre.hpp): k_pso_ring / k_pso_eval2 + k_pso_step passes
    size_t PB = 0;       // bytes of records per candidate
    int ringCUs = 0; // CUs' worth of resident waves the ring launch may take (a part of a streamed round: its share)
    Slice sl[16];
    int *cnt = nullptr, *nextCnt = nullptr;
    size_t SB = 0, EB = 0, WB = 0;
    pais_patch_result *d_out = nullptr;
    const pais_candidate *d_cands = nullptr;
    Timed tp;
};

struct pais_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int numCUs = 256;
    size_t ldsLimit = 64 * 1024;
    DevScene sc;
    DevCamera *d_cams = nullptr;
    uint8_t *d_img = nullptr;
    PaisImgT *d_imgF = nullptr;         // tap copy of d_img (element offsets identical; pais_internal.h PAIS_IMG_MODE)
    double *d_edge = nullptr;
    bool edgesOnTheFly = false;
    double *d_gauss = nullptr;
    size_t imgBytes = 0, edgeBytes = 0;
    // work buffers (grown on demand, never shrunk)
    pais_candidate *d_cands = nullptr;
    pais_patch_result *d_recs = nullptr;
    size_t recCap = 0;
    double *d_hp = nullptr;
    size_t hpBytes = 0;
    double *d_ratios = nullptr;         // region ratio per (candidate, visible camera) (k_region_ratio)
    size_t ratioBytes = 0;
    int *d_counters = nullptr;          // two sets of 4: [1] "needs another pass" count, [2] active-list length; pass p uses set
                                        // passSerial & 1, its k_after<2> clears the other one for the pass that follows
    unsigned passSerial = 0;
    bool countersDirty = false;         // a batch ended in an error: both sets are cleared before the next one
    pais_candidate *h_cands = nullptr;  // pinned staging of pais_refine_batch
    pais_patch_result *h_recs = nullptr;
    unsigned char *d_evalBlocks = nullptr; // per candidate: EvalPatch + EvalCam[M] (pais_eval.hpp), written by k_pso_init
    size_t evalBlockCap = 0;
    unsigned char *d_win = nullptr;     // per candidate: the reference window WinPix[S*S] of the PSO run
    size_t winCap = 0;
    int *d_active = nullptr;            // compacted indices of the candidates that run a PSO in the current pass
    size_t activeCap = 0;
    unsigned long long *d_stat = nullptr; // [0] evals [1] evals*bytesPerPixel [2] patches [3] ncc tables [4] tables*K
    int *h_counters = nullptr;          // pinned
    unsigned char *d_psoStates = nullptr; // one PsoState block per candidate
    size_t psoStateBytes = 0;
    // PSO pass of a batch: small batches run one k_pso_iter launch per iteration (the step is replayed
    // inside the evaluation waves); batches of at least `splitAbove` evaluation waves are throughput bound and run
    // k_pso_eval2 + k_pso_step per iteration, cut into slices whose sequences overlap on separate HIP streams
    long splitAbove = 0;
    int psoMinPer = 64;
    int evalParts = 0;                  // waves per cost evaluation in k_pso_iter (1, 2, 4); 0 = chosen per slice
    double partFill = 0.5;              // ... such that parts * waves <= partFill * numCUs * 16 (= the 2 waves per SIMD that the multi-wave kernels'
                                        // 192 registers allow: they are built without an occupancy target, pais_kernels.hip PAIS_ITER_BOUNDS)
    int psoStreams = 2;
    int tileStrip2 = 14, tileStrip1 = 24; // 64-pixel steps per strip of the two instantiations (PAIS_TILE_STRIP2 / PAIS_TILE_STRIP1)
    int arithLiteral = 0;               // PAIS_ARITH=literal: the cost in the reference's own statements and summation order (pais_literal.hpp);
                                        // every batch then runs the launch-per-iteration pipeline k_pso_eval_lit + k_pso_step
    int tileSplit = 1;                  // PAIS_TILE_SPLIT: 1 (default) the sixteen-wave kernel of pais_tile2.hpp for every tile batch (a particle's cameras
                                        // shared by two waves: <= 128 VGPRs, 4 waves / SIMD), k > 1 for batches of >= k cameras, 0 k_pso_tile.  Same bits.
    int tileStripSplit = 16, tileBias = 3; // PAIS_TILE_STRIP_SPLIT / PAIS_TILE_BIAS: its strip length; cameras the first half takes beyond an even share
    int tileForceNs1 = 0;               // PAIS_TILE_FORCE_NS1 (tests): the one-pixel instantiation also for batches of <= 32 cameras
    unsigned char *d_tileH = nullptr;   // homography scratch of the tile kernel (pais_tile.hpp PAIS_TILE_SCALAR_H): 16 regions of tileHSlice bytes
    size_t tileHBytes = 0, tileHSlice = 0;
    bool tileVerify = false;            // PAIS_TILE_VERIFY=1: every particle is ALSO walked by k_pso_eval2 and the two values compared (diagnosis)
    bool tileDebug = false;             // PAIS_TILE_DEBUG=1: counters of the tile kernel (printed by pais_get_kernel_stats)
    int tileMode = 1;                   // PAIS_TILE: 0 many-camera batches keep the one-wave-per-evaluation kernels; 1 the tile kernel for
                                        // scenes that tap the byte blob; 2 for every scene
    long tileAbove = 512;               // PAIS_TILE_ABOVE: waves per iteration from which a tile-eligible batch runs the tile kernel
    std::vector<hipStream_t> sub;       // sub-streams
    std::vector<hipEvent_t> subDone;
    hipEvent_t forkEv = nullptr;
    // fitness batch buffers
    pais_patch_state *d_states = nullptr;
    int32_t *d_idx = nullptr;
    double *d_particles = nullptr, *d_out = nullptr;
    size_t stateCap = 0, evalCap = 0;
    // neighbour count buffers
    double *d_nbC = nullptr;
    int32_t *d_nbN = nullptr;
    size_t nbCap = 0;
    // timing: HIP events are recorded only while profiling is on (pais_ctx_set_fine_timing); every pair is returned to
    // evFree by pais_get_kernel_stats, so the number of live events is bounded by one instrumented batch
    bool fineTiming = false;
    std::vector<EventPair> evPso, evBegin, evAfter, evEval, evEval2;
    std::vector<EventPair> evFree;
    double psoMs = 0, beginMs = 0, afterMs = 0, evalMs = 0, eval2Ms = 0;
    hipEvent_t refEv = nullptr;         // common time origin of the evEval2 intervals (recorded when fine timing is switched on; a lane uses its parent's)
    std::vector<std::pair<float, float>> eval2Intervals; // [start, end] of the drained evEval2 pairs, ms since refEv
    int64_t psoLaunches = 0, evalLaunches = 0, eval2Launches = 0, tileLaunches = 0, ringLaunches = 0;
    // lanes (pais_ctx_fork_lane): contexts over this one's scene with their own stream and work buffers
    pais_ctx *parent = nullptr;         // != nullptr: this is a lane; the scene's allocations belong to the parent
    std::vector<pais_ctx *> lanes;
    int openBatch = -1;                 // records of the batch between pais_refine_batch_begin and _end (-1: none)
    // k_pso_ring (pais_kernels.hip): the PSO pass of a large expansion batch as ONE launch over a device-side task ring
    double ringPerCam = 1843.0;         // PAIS_RING_PER_CAM: 9216 waves per iteration at five cameras
    int ringMode = 1;                   // PAIS_PSO_RING=0: large batches take the per-iteration launches (k_pso_eval2 + k_pso_step) instead
    int preMode = 1;                    // PAIS_PRE_SETUP=0: every evaluation wave sets itself up (round 5 behaviour; A/B, tests)
    double *d_pre = nullptr;            // per candidate and particle {status, homographies} (pais_pre.hpp)
    size_t preBytes = 0;
    unsigned *d_ring = nullptr;         // task ring
    size_t ringBytes = 0;
    unsigned *d_ringCtl = nullptr;      // head, tail, done, error
    unsigned *h_ringCtl = nullptr;      // pinned copy, read when the batch ends
    int *d_arrive = nullptr;            // per candidate: evaluations of the current iteration that have been delivered
    size_t arriveBytes = 0;
    bool ringUsed = false;              // by the open batch (its error words are read when the batch ends)
    bool ringSuppressed = false;        // while a batch whose ring pass failed is re-run
    bool ringSuppressOnce = false;      // the next device batch is such a re-run (pais_ctx_batch_status returned 1)
    int ringPendingN = 0;               // > 0: candidates of the pais_refine_batch_device_async batch whose ring status has not been read yet
    double ringTimeoutMs = 10000.0;     // PAIS_RING_TIMEOUT_MS: longest wait of a k_pso_ring wave for a ring entry (wall time, s_memrealtime)
    long ringSeedAbove = 0;             // PAIS_RING_SEED_ABOVE: evaluation waves per iteration from which a pass of a SEED batch takes the ring
    size_t ringMaxBytes = (size_t)512 << 20; // PAIS_RING_MAX_MB: batches whose ring would be larger keep the per-iteration launches
    int64_t ringFallbacks = 0;          // ring passes that did not complete and were re-run through the per-iteration launches
    std::vector<EventPair> evRing;      // the ring launches alone (fine timing)
    double ringMs = 0;
    PassPlan plan;                      // of the open batch (pais_refine_batch_open .. _enqueue)
    bool planDone = true;               // the open batch is enqueued to its end
    int roundHint = 0;                  // pais_ctx_set_round_hint: candidates of the round the next batch is a part of (0: it is the round)
};

// MVS::initPatchDistanceWeighting, mvs.cpp:97-114 (host; same arithmetic as the reference)
static void build_gauss(const pais_config &cfg, std::vector<double> &g)
{
    const int S = cfg.patchSize, r = cfg.patchRadius;
    g.assign((size_t)S * S, 0.0);
    const double sigma = cfg.distWeighting;
    const double s2 = 1.0 / (2.0 * sigma * sigma);
    const double s = 1.0 / (2.0 * M_PI * sigma * sigma);
    for (int x = 0; x < S; ++x)
        for (int y = 0; y < S; ++y) {
            double e = -(pow((double)(x - r), 2) + pow((double)(y - r), 2)) * s2;
            g[(size_t)x * S + y] = s * exp(e);
        }
    double n = 0;
    for (size_t i = 0; i < g.size(); ++i) n += g[i];
    const double inv = 1.0 / n;
    for (size_t i = 0; i < g.size(); ++i) g[i] = g[i] * inv;
}

static int apply_config(pais_ctx *ctx, const pais_config *cfg)
{
    pais_config c = *cfg;
    c.patchSize = (c.patchRadius << 1) + 1; // mvs.cpp:67
    if (c.patchRadius < 1 || c.patchRadius > 127) return fail_msg("patchRadius out of range");
    if (c.particleNum < 1 || c.particleNum * 2 > PAIS_MAX_PARTICLES) return fail_msg("particleNum out of range");
    if (c.maxLOD < 0 || c.maxLOD >= PAIS_MAX_LEVELS) return fail_msg("maxLOD out of range");
    std::vector<double> g;
    build_gauss(c, g);
    if (ctx->d_gauss) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipFree(ctx->d_gauss));
        ctx->d_gauss = nullptr;
    }
    HIPCHK(hipMalloc(&ctx->d_gauss, g.size() * sizeof(double)));
    HIPCHK(hipMemcpy(ctx->d_gauss, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->sc.cfg = c;
    ctx->sc.gauss = ctx->d_gauss;
    for (int l = 0; l < PAIS_MAX_LEVELS; ++l) ctx->sc.lodScale[l] = pow(c.lodRatio, l);
    return 0;
}

extern "C" const char *pais_last_error(void) { return g_err.c_str(); }
extern "C" size_t pais_sizeof_config(void) { return sizeof(pais_config); }
extern "C" size_t pais_sizeof_camera_desc(void) { return sizeof(pais_camera_desc); }
extern "C" size_t pais_sizeof_candidate(void) { return sizeof(pais_candidate); }
extern "C" size_t pais_sizeof_patch_result(void) { return sizeof(pais_patch_result); }
extern "C" uint32_t pais_rand31(uint64_t seed, uint64_t key, uint32_t run, uint32_t k) { return pais::rand31(seed, key, run, k); }
extern "C" uint64_t pais_child_key(uint64_t parent_key, int cam, int cx, int cy) { return pais::child_key(parent_key, cam, cx, cy); }

static int ctx_init(pais_ctx *ctx, const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device, uint64_t pso_seed);
static int ctx_init_work(pais_ctx *ctx);
extern "C" int pais_ctx_create(const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device,
                               uint64_t pso_seed, pais_ctx **out)
{
    if (!cfg || !cams || !out || num_cams <= 0) return fail_msg("pais_ctx_create: bad argument");
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev <= 0) return fail_msg("pais_ctx_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail_msg("pais_ctx_create: bad device index");
    HIPCHK(hipSetDevice(device));
    pais_ctx *ctx = new pais_ctx();
    // every failure below -- a validation message or a HIP error, e.g. hipMalloc of a multi-GB blob -- leaves through
    // ctx_init's return value; what had been created by then is released here
    const int rc = ctx_init(ctx, cfg, num_cams, cams, device, pso_seed);
    if (rc) { pais_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return 0;
}

static int ctx_init(pais_ctx *ctx, const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device, uint64_t pso_seed)
{
    ctx->device = device;
    memset(&ctx->sc, 0, sizeof(ctx->sc));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ctx->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->ldsLimit = prop.sharedMemPerBlock > 0 ? (size_t)prop.sharedMemPerBlock : 64 * 1024;
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->sc.seed = pso_seed;
    ctx->sc.numCams = num_cams;
    int rc = apply_config(ctx, cfg);
    if (rc) return rc;
    const bool wantEdge = cfg->adaptiveGradientEnable != 0;
    // edge pyramids: as the caller built them (Camera::pyramidEdge, camera.cpp:72-77,87-91), or -- level_edge == NULL --
    // evaluated on the fly from the gray levels with the same statements (no double-precision copy of every pyramid)
    const bool edgesGiven = wantEdge && cams[0].level_edge[0] != nullptr;

    // layout the blobs
    std::vector<DevCamera> hc((size_t)num_cams);
    std::vector<size_t> imgOff((size_t)num_cams * PAIS_MAX_LEVELS, 0), edgeOff((size_t)num_cams * PAIS_MAX_LEVELS, 0);
    size_t imgBytes = 0, edgeBytes = 0;
    for (int c = 0; c < num_cams; ++c) {
        const pais_camera_desc &d = cams[c];
        if (d.max_lod < 0 || d.max_lod >= PAIS_MAX_LEVELS) return fail_msg("camera max_lod out of range");
        for (int l = 0; l <= d.max_lod; ++l) {
            if (!d.level_image[l] || d.level_width[l] <= 0 || d.level_height[l] <= 0) return fail_msg("camera level missing");
            // the evaluation packs a level's tap bounds (w - 4, h - 4) into 16 bits each and its row offsets into 24 (pais_eval.hpp)
            if (d.level_width[l] > 65535 || d.level_height[l] > 65535) return fail_msg("camera level larger than 65535 pixels in one dimension");
            size_t px = (size_t)d.level_width[l] * d.level_height[l];
            if (px >= ((size_t)1 << 29)) return fail_msg("camera level of 2^29 pixels or more (32-bit tap offsets inside a level)");
            imgOff[(size_t)c * PAIS_MAX_LEVELS + l] = imgBytes;
            imgBytes = (imgBytes + px + 16 + 255) & ~(size_t)255; // +16: taps read (px+1, py+1)
            if (wantEdge && edgesGiven) {
                if (!d.level_edge[l]) return fail_msg("level_edge given for some levels only");
                edgeOff[(size_t)c * PAIS_MAX_LEVELS + l] = edgeBytes;
                edgeBytes = (edgeBytes + px * sizeof(double) + 255) & ~(size_t)255;
            }
        }
    }
    // pyramid levels straight from the caller's buffers into HBM (rows repacked to stride == width); the padding between
    // levels is zero
    if (imgBytes >= ((size_t)1 << 40)) return fail_msg("pyramids of 2^40 bytes or more (a level's start is packed into 40 bits next to its homography)");
    HIPCHK(hipMalloc(&ctx->d_img, imgBytes));
    HIPCHK(hipMemsetAsync(ctx->d_img, 0, imgBytes, ctx->stream));
    for (int c = 0; c < num_cams; ++c) {
        const pais_camera_desc &d = cams[c];
        for (int l = 0; l <= d.max_lod; ++l) {
            const int w = d.level_width[l], h = d.level_height[l];
            const size_t stride = d.level_stride[l] > 0 ? (size_t)d.level_stride[l] : (size_t)w;
            HIPCHK(hipMemcpy2DAsync(ctx->d_img + imgOff[(size_t)c * PAIS_MAX_LEVELS + l], (size_t)w, d.level_image[l], stride, (size_t)w,
                                    (size_t)h, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ctx->imgBytes = imgBytes;
    // the float2 tap copy is expanded on the device -- for scenes whose pyramids are small enough for it to pay
    // (pais_internal.h PaisImgT); larger scenes tap the byte blob
    {
        size_t maxMB = 256;
        if (const char *e = getenv("PAIS_TAP_FLOAT_MAX_MB")) maxMB = (size_t)strtoull(e, nullptr, 10);
        if (imgBytes <= (maxMB << 20)) {
            HIPCHK(hipMalloc(&ctx->d_imgF, imgBytes * sizeof(PaisImgT)));
            HIPCHK(pais_launch::expand_image(ctx->d_img, ctx->d_imgF, imgBytes, ctx->stream));
        }
    }
    if (wantEdge && edgesGiven && edgeBytes) {
        HIPCHK(hipMalloc(&ctx->d_edge, edgeBytes));
        for (int c = 0; c < num_cams; ++c) {
            const pais_camera_desc &d = cams[c];
            for (int l = 0; l <= d.max_lod; ++l) {
                size_t px = (size_t)d.level_width[l] * d.level_height[l];
                HIPCHK(hipMemcpyAsync((uint8_t *)ctx->d_edge + edgeOff[(size_t)c * PAIS_MAX_LEVELS + l], d.level_edge[l],
                                      px * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            }
        }
        ctx->edgeBytes = edgeBytes;
    }
    // edge maps on the fly (level_edge == NULL): only the per-level minimum / maximum magnitude is precomputed
    std::vector<unsigned long long> mm;
    if (wantEdge && !edgesGiven) {
        mm.assign((size_t)num_cams * PAIS_MAX_LEVELS * 2, 0ULL);
        for (size_t i = 0; i < mm.size(); i += 2) mm[i] = ~0ULL;
        struct DevBuf { unsigned long long *p = nullptr; ~DevBuf() { (void)hipFree(p); } } mmBuf; // released on every path
        HIPCHK(hipMalloc(&mmBuf.p, mm.size() * sizeof(unsigned long long)));
        unsigned long long *d_mm = mmBuf.p;
        HIPCHK(hipMemcpyAsync(d_mm, mm.data(), mm.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
        for (int c = 0; c < num_cams; ++c)
            for (int l = 0; l <= cams[c].max_lod; ++l)
                HIPCHK(pais_launch::level_edge_minmax(ctx->d_img + imgOff[(size_t)c * PAIS_MAX_LEVELS + l], cams[c].level_width[l],
                                                      cams[c].level_height[l], d_mm + ((size_t)c * PAIS_MAX_LEVELS + l) * 2, ctx->stream));
        HIPCHK(hipMemcpyAsync(mm.data(), d_mm, mm.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->edgesOnTheFly = wantEdge && !edgesGiven;
    for (int c = 0; c < num_cams; ++c) {
        const pais_camera_desc &d = cams[c];
        DevCamera &h = hc[c];
        memset(&h, 0, sizeof(h));
        memcpy(h.KR, d.KR, sizeof(h.KR));
        memcpy(h.KT, d.KT, sizeof(h.KT));
        memcpy(h.R, d.rotation, sizeof(h.R));
        memcpy(h.T, d.translation, sizeof(h.T));
        memcpy(h.C, d.center, sizeof(h.C));
        memcpy(h.optN, d.optical_normal, sizeof(h.optN));
        h.focal[0] = d.focal[0]; h.focal[1] = d.focal[1];
        h.pp[0] = d.principle_point[0]; h.pp[1] = d.principle_point[1];
        h.maxLOD = d.max_lod;
        for (int l = 0; l <= d.max_lod; ++l) {
            h.w[l] = d.level_width[l];
            h.h[l] = d.level_height[l];
            h.imgOff[l] = (uint64_t)imgOff[(size_t)c * PAIS_MAX_LEVELS + l];
            h.edgeOff[l] = (uint64_t)(edgeOff[(size_t)c * PAIS_MAX_LEVELS + l] / sizeof(double));
            if (!mm.empty()) {
                const unsigned long long lo = mm[((size_t)c * PAIS_MAX_LEVELS + l) * 2], hi = mm[((size_t)c * PAIS_MAX_LEVELS + l) * 2 + 1];
                memcpy(&h.edgeMin[l], &lo, 8);
                memcpy(&h.edgeMax[l], &hi, 8);
            }
        }
    }
    HIPCHK(hipMalloc(&ctx->d_cams, sizeof(DevCamera) * (size_t)num_cams));
    HIPCHK(hipMemcpy(ctx->d_cams, hc.data(), sizeof(DevCamera) * (size_t)num_cams, hipMemcpyHostToDevice));
    ctx->sc.cams = ctx->d_cams;
    ctx->sc.imgBlob = ctx->d_img;
    ctx->sc.imgF = ctx->d_imgF;
    ctx->sc.edgeBlob = ctx->d_edge;

    return ctx_init_work(ctx);
}

// everything of a context that is not the scene: counters, statistics, knobs, sub-streams (a lane has its own)
static int ctx_init_work(pais_ctx *ctx)
{
    HIPCHK(hipMalloc(&ctx->d_counters, sizeof(int) * 8));
    HIPCHK(hipMemset(ctx->d_counters, 0, sizeof(int) * 8));
    HIPCHK(hipMalloc(&ctx->d_stat, sizeof(unsigned long long) * 24));
    HIPCHK(hipMemset(ctx->d_stat, 0, sizeof(unsigned long long) * 24));
    HIPCHK(hipHostMalloc((void **)&ctx->h_counters, sizeof(int) * 4, hipHostMallocDefault));
    if (const char *e = getenv("PAIS_EVAL_PARTS")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) ctx->evalParts = v; }
    if (const char *e = getenv("PAIS_PART_FILL")) { double v = atof(e); if (v > 0) ctx->partFill = v; }
    ctx->splitAbove = 3L * ctx->numCUs * 12;
    if (const char *e = getenv("PAIS_SPLIT_ABOVE")) { long v = atol(e); if (v > 0) ctx->splitAbove = v; }
    if (const char *e = getenv("PAIS_PSO_MINPER")) { int v = atoi(e); if (v >= 1) ctx->psoMinPer = v; }
    if (const char *e = getenv("PAIS_PSO_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= 16) ctx->psoStreams = v; }
    if (const char *e = getenv("PAIS_PSO_RING")) ctx->ringMode = atoi(e);
    if (const char *e = getenv("PAIS_PRE_SETUP")) ctx->preMode = atoi(e);
    if (const char *e = getenv("PAIS_RING_PER_CAM")) ctx->ringPerCam = atof(e);
    if (const char *e = getenv("PAIS_RING_TIMEOUT_MS")) {
        // 0 is the tests' hook (every ring pass "times out" and is re-run launch by launch); anything else must be a finite wait
        // of at least a millisecond and at most ten minutes -- negative / NaN values are ignored
        const double v = atof(e);
        if (v == 0.0 && e[0] == '0') ctx->ringTimeoutMs = 0.0;
        else if (v >= 1.0 && v <= 600000.0) ctx->ringTimeoutMs = v;
        else fprintf(stderr, "pais: PAIS_RING_TIMEOUT_MS=%s ignored (0, or 1 .. 600000 ms)\n", e);
    }
    // (0 = never, the default: measured on the pawn bench, profiles/r04_seed_ring_ab.txt -- the first pass of the 200 seeds as one
    //  ring launch costs +3.5 ms per reconstruction against k_pso_iter's 2 x 62 launches: with 2N = 30 particles the one-wave
    //  swarm step is a long serial chain between two evaluations of a candidate, and 6000 tasks per iteration are two residency
    //  passes, so the chain of 61 iterations -- not the throughput -- bounds the launch)
    ctx->ringSeedAbove = 0;
    if (const char *e = getenv("PAIS_RING_SEED_ABOVE")) ctx->ringSeedAbove = atol(e);
    if (const char *e = getenv("PAIS_RING_MAX_MB")) { long v = atol(e); if (v > 0) ctx->ringMaxBytes = (size_t)v << 20; }
    HIPCHK(hipMalloc(&ctx->d_ringCtl, PAIS_RING_CTL_BYTES * PAIS_RINGS));
    HIPCHK(hipMemset(ctx->d_ringCtl, 0, PAIS_RING_CTL_BYTES * PAIS_RINGS));
    HIPCHK(hipHostMalloc((void **)&ctx->h_ringCtl, PAIS_RING_CTL_BYTES * PAIS_RINGS, hipHostMallocDefault));
    memset(ctx->h_ringCtl, 0, PAIS_RING_CTL_BYTES * PAIS_RINGS);
    if (const char *e = getenv("PAIS_TILE")) ctx->tileMode = atoi(e);
    if (const char *e = getenv("PAIS_TILE_DEBUG")) ctx->tileDebug = atoi(e) != 0;
    if (const char *e = getenv("PAIS_TILE_VERIFY")) ctx->tileVerify = atoi(e) != 0;
    if (const char *e = getenv("PAIS_TILE_STRIP2")) { int v = atoi(e); if (v >= 2) ctx->tileStrip2 = v; }
    if (const char *e = getenv("PAIS_TILE_STRIP1")) { int v = atoi(e); if (v >= 1) ctx->tileStrip1 = v; }
    if (const char *e = getenv("PAIS_TILE_FORCE_NS1")) ctx->tileForceNs1 = atoi(e) != 0;
    if (const char *e = getenv("PAIS_ARITH")) ctx->arithLiteral = (strcmp(e, "literal") == 0);
    if (const char *e = getenv("PAIS_TILE_SPLIT")) ctx->tileSplit = atoi(e) > 0 ? atoi(e) : 0; // 0 off, 1 every batch, k > 1: batches of >= k cameras
    if (const char *e = getenv("PAIS_TILE_STRIP_SPLIT")) { int v = atoi(e); if (v >= 1) ctx->tileStripSplit = v; }
    if (const char *e = getenv("PAIS_TILE_BIAS")) ctx->tileBias = atoi(e);
    if (const char *e = getenv("PAIS_TILE_ABOVE")) { long v = atol(e); if (v > 0) ctx->tileAbove = v; }
    HIPCHK(hipEventCreateWithFlags(&ctx->forkEv, hipEventDisableTiming));
    for (int i = 0; i + 1 < ctx->psoStreams; ++i) { // slice 0 runs on ctx->stream itself
        hipStream_t st; hipEvent_t ev;
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->sub.push_back(st);
        ctx->subDone.push_back(ev);
    }
    if (const char *e = getenv("PAIS_FINE_TIMING")) ctx->fineTiming = atoi(e) != 0;
    return 0;
}

extern "C" void pais_ctx_destroy(pais_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->parent) { // a lane destroyed on its own: the parent forgets it
        auto &v = ctx->parent->lanes;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == ctx) { v.erase(v.begin() + (long)i); break; }
        ctx->parent = nullptr;
    }
    while (!ctx->lanes.empty()) {
        pais_ctx *l = ctx->lanes.back();
        ctx->lanes.pop_back();
        l->parent = nullptr;
        pais_ctx_destroy(l);
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    auto freeEv = [](std::vector<EventPair> &v) {
        for (auto &p : v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        v.clear();
    };
    freeEv(ctx->evPso); freeEv(ctx->evBegin); freeEv(ctx->evAfter); freeEv(ctx->evEval); freeEv(ctx->evEval2); freeEv(ctx->evFree);
    (void)hipFree(ctx->d_psoStates);
    (void)hipFree(ctx->d_win);
    (void)hipFree(ctx->d_ratios);
    (void)hipFree(ctx->d_nbC); (void)hipFree(ctx->d_nbN);
    for (auto st : ctx->sub) (void)hipStreamDestroy(st);
    for (auto ev : ctx->subDone) (void)hipEventDestroy(ev);
    if (ctx->forkEv) (void)hipEventDestroy(ctx->forkEv);
    if (ctx->refEv) (void)hipEventDestroy(ctx->refEv);
    pais_launch::ring_profile_print();
    (void)hipFree(ctx->d_pre);
    (void)hipFree(ctx->d_ring); (void)hipFree(ctx->d_ringCtl); (void)hipFree(ctx->d_arrive); (void)hipFree(ctx->d_tileH);
    if (ctx->h_ringCtl) (void)hipHostFree(ctx->h_ringCtl);
    (void)hipFree(ctx->d_cams); (void)hipFree(ctx->d_img); (void)hipFree(ctx->d_imgF); (void)hipFree(ctx->d_edge); (void)hipFree(ctx->d_gauss);
    (void)hipFree(ctx->d_cands); (void)hipFree(ctx->d_recs); (void)hipFree(ctx->d_hp);
    (void)hipFree(ctx->d_counters); (void)hipFree(ctx->d_stat); (void)hipFree(ctx->d_active); (void)hipFree(ctx->d_evalBlocks); (void)hipHostFree(ctx->h_cands); (void)hipHostFree(ctx->h_recs);
    (void)hipFree(ctx->d_states); (void)hipFree(ctx->d_idx); (void)hipFree(ctx->d_particles); (void)hipFree(ctx->d_out);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pais_ctx_set_config(pais_ctx *ctx, const pais_config *cfg)
{
    if (!ctx || !cfg) return fail_msg("pais_ctx_set_config: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (cfg->adaptiveGradientEnable && !ctx->d_edge && !ctx->edgesOnTheFly)
        return fail_msg("adaptiveGradientEnable needs the gradient weighting enabled at create time (edge pyramids or their on-the-fly statistics)");
    if (ctx->parent) return fail_msg("pais_ctx_set_config: a lane takes its configuration from its parent");
    for (pais_ctx *l : ctx->lanes) HIPCHK(hipStreamSynchronize(l->stream)); // (the table is reallocated)
    int rc = apply_config(ctx, cfg);
    if (rc) return rc;
    for (pais_ctx *l : ctx->lanes) {
        l->sc.cfg = ctx->sc.cfg;
        l->sc.gauss = ctx->sc.gauss;
        for (int i = 0; i < PAIS_MAX_LEVELS; ++i) l->sc.lodScale[i] = ctx->sc.lodScale[i];
    }
    return 0;
}

extern "C" int pais_ctx_fork_lane(pais_ctx *parent, pais_ctx **out)
{
    if (!parent || !out) return fail_msg("pais_ctx_fork_lane: bad argument");
    if (parent->parent) return fail_msg("pais_ctx_fork_lane: a lane cannot be forked");
    HIPCHK(hipSetDevice(parent->device));
    pais_ctx *l = new pais_ctx();
    l->device = parent->device;
    l->numCUs = parent->numCUs;
    l->ldsLimit = parent->ldsLimit;
    l->sc = parent->sc; // the parent's cameras, blobs and table: shared, not owned (d_cams, d_img ... stay null here)
    l->edgesOnTheFly = parent->edgesOnTheFly;
    l->imgBytes = parent->imgBytes;
    l->edgeBytes = parent->edgeBytes;
    hipError_t e = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete l; return fail("hipStreamCreateWithFlags", e); }
    const int rc = ctx_init_work(l);
    if (rc) { pais_ctx_destroy(l); return rc; }
    l->fineTiming = parent->fineTiming;
    l->parent = parent;
    parent->lanes.push_back(l);
    *out = l;
    return 0;
}

extern "C" int pais_ctx_set_neighbor_radius(pais_ctx *ctx, double r)
{
    if (!ctx) return fail_msg("bad ctx");
    ctx->sc.cfg.neighborRadius = r;
    for (pais_ctx *l : ctx->lanes) l->sc.cfg.neighborRadius = r;
    return 0;
}

extern "C" void *pais_ctx_stream(pais_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int pais_ctx_synchronize(pais_ctx *ctx)
{
    if (!ctx) return fail_msg("bad ctx");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ------------------------------------------------------------------- timing --
static int get_event_pair(pais_ctx *ctx, EventPair &p)
{
    if (!ctx->evFree.empty()) {
        p = ctx->evFree.back();
        ctx->evFree.pop_back();
        return 0;
    }
    HIPCHK(hipEventCreate(&p.a));
    HIPCHK(hipEventCreate(&p.b));
    return 0;
}
static int drain_events(pais_ctx *ctx, std::vector<EventPair> &v, double &acc)
{
    for (auto &p : v) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
        acc += ms;
        ctx->evFree.push_back(p);
    }
    v.clear();
    return 0;
}
int Timed::begin(pais_ctx *c, hipStream_t s, std::vector<EventPair> *v)
{
    ctx = c; st = s; into = v; on = c->fineTiming;
    if (!on) return 0;
    if (get_event_pair(c, e)) return -2;
    HIPCHK(hipEventRecord(e.a, s));
    return 0;
}
int Timed::end()
{
    if (!on) return 0;
    HIPCHK(hipEventRecord(e.b, st));
    into->push_back(e);
    return 0;
}

template <typename T> static int grow(pais_ctx *ctx, T *&ptr, size_t &capBytes, size_t needBytes)
{
    if (needBytes <= capBytes) return 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ptr);
    ptr = nullptr;
    capBytes = 0;
    const size_t want = needBytes + needBytes / 2 + 4096;
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, want));
    ptr = (T *)q;
    capBytes = want;
    return 0;
}

// MVS::neighborPatchFiltering's distance counts (mvs.cpp:448-524) -- see k_neighbor_count
extern "C" int pais_neighbor_count(pais_ctx *ctx, int n, const double *centers, double radius, int32_t *counts, double *kernel_ms)
{
    if (!ctx || n < 0 || (n && (!centers || !counts))) return fail_msg("pais_neighbor_count: bad argument");
    if (kernel_ms) *kernel_ms = 0;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    if ((size_t)n > ctx->nbCap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_nbC); (void)hipFree(ctx->d_nbN);
        ctx->d_nbC = nullptr; ctx->d_nbN = nullptr; ctx->nbCap = 0;
        const size_t cap = (size_t)n + (size_t)n / 2 + 256;
        HIPCHK(hipMalloc(&ctx->d_nbC, sizeof(double) * 3 * cap));
        HIPCHK(hipMalloc(&ctx->d_nbN, sizeof(int32_t) * cap));
        ctx->nbCap = cap;
    }
    EventPair e{nullptr, nullptr};
    if (get_event_pair(ctx, e)) return -2;
    HIPCHK(hipMemcpyAsync(ctx->d_nbC, centers, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(e.a, ctx->stream));
    HIPCHK(pais_launch::neighbor_count(ctx->d_nbC, n, radius, ctx->d_nbN, ctx->stream));
    HIPCHK(hipEventRecord(e.b, ctx->stream));
    HIPCHK(hipMemcpyAsync(counts, ctx->d_nbN, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) *kernel_ms = ms;
    }
    ctx->evFree.push_back(e);
    return 0;
}

extern "C" int pais_ctx_set_fine_timing(pais_ctx *ctx, int on)
{
    if (!ctx) return fail_msg("pais_ctx_set_fine_timing: bad argument");
    if (on && !ctx->parent) {
        HIPCHK(hipSetDevice(ctx->device));
        if (!ctx->refEv) HIPCHK(hipEventCreate(&ctx->refEv));
        HIPCHK(hipEventRecord(ctx->refEv, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->fineTiming = on != 0;
    for (pais_ctx *l : ctx->lanes) l->fineTiming = on != 0;
    return 0;
}

// ------------------------------------------------------------ fitness batch --
extern "C" int pais_fitness_batch(pais_ctx *ctx, int n_states, const pais_patch_state *states, int n_evals,
                                  const int32_t *state_index, const double *particles, double *out)
{
    if (!ctx || n_states < 0 || n_evals < 0) return fail_msg("pais_fitness_batch: bad argument");
    if (n_evals == 0) return 0;
    if (!states || !state_index || !particles || !out) return fail_msg("pais_fitness_batch: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    int Kmax = 1;
    for (int i = 0; i < n_states; ++i) {
        const pais_patch_state &s = states[i];
        if (s.num_cam < 1 || s.num_cam > PAIS_MAX_VIS) return fail_msg("pais_fitness_batch: num_cam out of range");
        if (s.ref_cam < 0 || s.ref_cam >= ctx->sc.numCams || s.lod < 0 || s.lod >= PAIS_MAX_LEVELS) return fail_msg("pais_fitness_batch: bad ref_cam/lod");
        for (int k = 0; k < s.num_cam; ++k)
            if (s.cam_idx[k] < 0 || s.cam_idx[k] >= ctx->sc.numCams) return fail_msg("pais_fitness_batch: bad cam_idx");
        if (s.num_cam > Kmax) Kmax = s.num_cam;
    }
    for (int e = 0; e < n_evals; ++e)
        if (state_index[e] < 0 || state_index[e] >= n_states) return fail_msg("pais_fitness_batch: bad state_index");
    if ((size_t)n_states > ctx->stateCap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_states);
        ctx->d_states = nullptr;
        ctx->stateCap = (size_t)n_states * 2;
        HIPCHK(hipMalloc(&ctx->d_states, sizeof(pais_patch_state) * ctx->stateCap));
    }
    if ((size_t)n_evals > ctx->evalCap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_idx); (void)hipFree(ctx->d_particles); (void)hipFree(ctx->d_out);
        ctx->d_idx = nullptr; ctx->d_particles = nullptr; ctx->d_out = nullptr;
        ctx->evalCap = (size_t)n_evals * 2;
        HIPCHK(hipMalloc(&ctx->d_idx, sizeof(int32_t) * ctx->evalCap));
        HIPCHK(hipMalloc(&ctx->d_particles, sizeof(double) * 3 * ctx->evalCap));
        HIPCHK(hipMalloc(&ctx->d_out, sizeof(double) * ctx->evalCap));
    }
    if (grow(ctx, ctx->d_evalBlocks, ctx->evalBlockCap, pais_launch::eval_block_bytes_host(Kmax) * (size_t)n_states)) return -2;
    if (grow(ctx, ctx->d_win, ctx->winCap, pais_launch::win_bytes_per_candidate(ctx->sc) * (size_t)n_states)) return -2;
    HIPCHK(hipMemcpyAsync(ctx->d_states, states, sizeof(pais_patch_state) * (size_t)n_states, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_idx, state_index, sizeof(int32_t) * (size_t)n_evals, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_particles, particles, sizeof(double) * 3 * (size_t)n_evals, hipMemcpyHostToDevice, ctx->stream));
    Timed tf;
    if (tf.begin(ctx, ctx->stream, &ctx->evEval)) return -2; // kernel-only time, reported as eval_ms / eval_launches
    HIPCHK(pais_launch::fitness(ctx->sc, ctx->d_states, n_states, ctx->d_idx, ctx->d_particles, ctx->d_out, n_evals, Kmax,
                                ctx->d_evalBlocks, ctx->d_win, ctx->arithLiteral, ctx->stream));
    if (tf.end()) return -2;
    ctx->evalLaunches++;
    HIPCHK(hipMemcpyAsync(out, ctx->d_out, sizeof(double) * (size_t)n_evals, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ------------------------------------------------------------- refine batch --
// One PSO pass of a batch as the host enqueues it: opened (pipeline chosen, slices forked to the sub-streams), its
// iterations enqueued in one or several pieces, closed (sub-streams joined, after-stage).  Pieces exist so that the
// launch sequences of two lanes can be interleaved by a driver (pais_refine_batch_open / _enqueue).
static int pass_open(pais_ctx *ctx, PassPlan &P, int pass, int againCount)
{
    const DevScene &sc = ctx->sc;
    P.cnt = ctx->d_counters + 4 * (ctx->passSerial & 1);
    P.nextCnt = ctx->d_counters + 4 * ((ctx->passSerial + 1) & 1);
    ctx->passSerial++;
    if (P.tp.begin(ctx, ctx->stream, &ctx->evPso)) return -2;
    if (pass > 0) // (pass 0: done by the begin launch)
        HIPCHK(pais_launch::pso_init(sc, P.d_out, P.n, ctx->d_psoStates, P.Nmax, ctx->d_active, P.cnt + 2, ctx->d_evalBlocks, ctx->d_win, P.Kmax,
                                     ctx->stream));
    // k_pso_iter needs the swarm of a candidate in the lanes of one wave; a batch of several residency passes
    // (>= 3 x 12 waves per CU) is throughput bound: there the step replay in every evaluation wave (~13 % of a wave's
    // time) costs more than a separate one-wave-per-candidate k_pso_step launch per iteration, whose latency the other
    // sub-stream hides
    // the tile kernel pays where the taps miss the caches: scenes whose pyramids exceed PAIS_TAP_FLOAT_MAX_MB (those that tap
    // the byte blob; the dome: PSO passes -22 %).  On the 184 MB ring, whose taps hit L2, it LOSES 35 % against the
    // one-wave kernels (profiles/r03_ring_tile_ab.txt) -- PAIS_TILE=2 forces it regardless (tests)
    const bool tileOk = (ctx->tileMode == 2 || (ctx->tileMode == 1 && ctx->sc.imgF == nullptr)) && pais_launch::tile_eligible(P.Kmax);
    // (the tile kernel is an evaluation launch of the large-batch pipeline: batches that are eligible for it take that
    // pipeline from PAIS_TILE_ABOVE waves per iteration on)
    // (a batch that is one part of a streamed round shares the GPU with the other part: the pipeline is chosen for the round)
    const int n = P.n, Nmax = P.Nmax;
    const int nPlan = ctx->roundHint > n ? ctx->roundHint : n;
    P.useIter = Nmax <= 64 && (long)nPlan * Nmax < (tileOk ? ctx->tileAbove : ctx->splitAbove);
    P.useTile = tileOk && !P.useIter;
    if (ctx->arithLiteral) P.useIter = P.useTile = false; // (and no ring below): k_pso_eval_lit + k_pso_step for every batch
    // the PSO pass as ONE launch over device-side task rings (k_pso_ring): host batches, device-pointer batches (the shards of
    // the multi-GPU drivers) and the passes of seed batches alike -- the error words are read at the batch's (pass's) next
    // synchronisation point and a pass that did not complete is re-run through the per-iteration launches (ring_failed below).
    // (a candidate's chain of maxIt evaluation + step latencies bounds a ring launch from below; the longer an evaluation -- more
    // cameras --, the more candidates it takes for throughput to dominate that chain: PAIS_RING_PER_CAM waves per iteration and camera)
    const long ringWaves = (long)(P.hasSeeds && pass > 0 ? againCount : n) * Nmax;
    const size_t ringNeed = pais_launch::ring_words(n, Nmax, P.maxIt) * sizeof(unsigned);
    P.useRing = ctx->ringMode != 0 && !ctx->arithLiteral && !ctx->ringSuppressed && !P.useTile && Nmax <= 64 && n < (1 << 24) && ringNeed <= ctx->ringMaxBytes &&
                (P.hasSeeds ? (ctx->ringSeedAbove > 0 && ringWaves >= ctx->ringSeedAbove)
                            : (!P.useIter && ringWaves >= (long)(ctx->ringPerCam * P.Kmax)));
    // (the parts of a streamed round keep the per-iteration launches, which interleave on their lanes; two ring launches would
    // run one after the other, or each on its share of the CUs -- ring scene 17.5 s against 18.3 / 19.7 s that way)
    if (P.useRing && nPlan > n && ctx->ringMode != 3) P.useRing = false;
    if (P.useRing) P.useIter = false;
    P.ringCUs = ctx->numCUs;
    if (P.useRing) {
        if (grow(ctx, ctx->d_ring, ctx->ringBytes, ringNeed)) return -2;
        if (grow(ctx, ctx->d_arrive, ctx->arriveBytes, sizeof(int) * PAIS_ARRIVE_STRIDE * (size_t)n)) return -2;
    }
    // the large-batch pipelines of the kernel arithmetic read the evaluations' set-up from records the swarm step writes
    P.usePre = ctx->preMode != 0 && !ctx->arithLiteral && !P.useIter && !P.useTile && !(P.useRing && !pais_launch::pre_ring_ok(P.Kmax));
    P.PB = pais_launch::pre_bytes_per_candidate(Nmax, P.Kmax);
    if (P.usePre) {
        if (grow(ctx, ctx->d_pre, ctx->preBytes, P.PB * (size_t)n)) return -2;
        HIPCHK(pais_launch::pso_setup0(sc, ctx->d_psoStates, n, Nmax, P.Kmax, ctx->d_evalBlocks, ctx->d_pre, ctx->stream));
    }
    if (P.useTile) { // homography scratch of the tile kernel's waves: a region per slice, 1024 workgroups' worth each
        const size_t slice = sizeof(double) * 10 * (size_t)PAIS_MAX_VIS * 8 * (PAIS_TILE_SCALAR_H ? 1024 : 1); // (unused unless the variant is built)
        if (grow(ctx, ctx->d_tileH, ctx->tileHBytes, slice * 16)) return -2;
        ctx->tileHSlice = slice;
    }
    // k_pso_iter works on the compacted list of candidates that run a PSO in this pass (k_pso_init);
    // its length is n at most in the first pass and exactly the "again" count afterwards
    const int nRun = P.useIter ? (pass == 0 ? n : againCount) : n;
    int S = P.useRing ? 1 : ctx->psoStreams;
    if (nPlan > n) S = (int)((double)S * n / nPlan + 0.5); // the round's sub-streams are shared out among its parts
    const int minPer = ctx->psoMinPer; // slices smaller than this only add launch overhead
    if (nRun < S * minPer) S = (nRun + minPer - 1) / minPer;
    if (S < 1) S = 1;
    P.S = S;
    if (S > 1) HIPCHK(hipEventRecord(ctx->forkEv, ctx->stream));
    // slice 0 stays on the context's own stream, the others fork to sub-streams and join back
    P.nSl = 0;
    for (int sI = 0; sI < S; ++sI) {
        PassPlan::Slice q;
        q.lo = (int)((long)nRun * sI / S);
        q.hi = (int)((long)nRun * (sI + 1) / S);
        if (q.hi <= q.lo) continue;
        q.own = (S == 1) || (sI == 0);
        q.st = q.own ? ctx->stream : ctx->sub[sI - 1];
        if (!q.own) HIPCHK(hipStreamWaitEvent(q.st, ctx->forkEv, 0));
        // waves per evaluation: a slice that leaves the GPU mostly empty is bound by the latency of
        // one evaluation wave, so share each evaluation among 4 (2) waves while they all stay resident
        q.parts = 1;
        if (P.useIter) {
            const long waves = (long)(q.hi - q.lo) * Nmax, resident = (long)((double)ctx->numCUs * 16 * ctx->partFill * n / nPlan);
            q.parts = ctx->evalParts > 0 ? ctx->evalParts : (waves * 4 <= resident ? 4 : (waves * 2 <= resident ? 2 : 1));
        }
        P.sl[P.nSl++] = q;
    }
    P.itNext = 0;
    return 0;
}

// iterations [P.itNext, itEnd) of the pass (itEnd is clamped to maxIt + 1: launch `it` = step it - 1 + cost of the moved swarm)
static int pass_iterations(pais_ctx *ctx, PassPlan &P, int itEnd)
{
    const DevScene &sc = ctx->sc;
    if (itEnd > P.maxIt + 1) itEnd = P.maxIt + 1;
    if (P.useRing) {
        if (P.itNext == 0 && itEnd > 0) {
            const unsigned long long ticks = (unsigned long long)(ctx->ringTimeoutMs * 1e5); // s_memrealtime counts at 100 MHz
            HIPCHK(pais_launch::pso_ring(sc, P.d_out, ctx->d_psoStates, P.n, P.Nmax, P.Kmax, P.maxIt, ctx->d_evalBlocks, ctx->d_win, ctx->d_ring,
                                         ctx->d_ringCtl, ctx->d_arrive, ctx->d_stat, P.ringCUs, 0, ticks, ctx->stream, P.usePre ? ctx->d_pre : nullptr));
            Timed te; // (the kernel alone)
            if (te.begin(ctx, ctx->stream, &ctx->evRing)) return -2;
            HIPCHK(pais_launch::pso_ring(sc, P.d_out, ctx->d_psoStates, P.n, P.Nmax, P.Kmax, P.maxIt, ctx->d_evalBlocks, ctx->d_win, ctx->d_ring,
                                         ctx->d_ringCtl, ctx->d_arrive, ctx->d_stat, P.ringCUs, 1, ticks, ctx->stream, P.usePre ? ctx->d_pre : nullptr));
            if (te.end()) return -2;
            HIPCHK(hipMemcpyAsync(ctx->h_ringCtl, ctx->d_ringCtl, PAIS_RING_CTL_BYTES * PAIS_RINGS, hipMemcpyDeviceToHost, ctx->stream));
            ctx->ringUsed = true;
            ctx->evalLaunches++;
            ctx->eval2Launches++;
            ctx->ringLaunches++;
            P.itNext = P.maxIt + 1; // the launch covers every iteration of the pass
        }
        return 0;
    }
    // enqueued iteration by iteration across the slices: every sub-stream has work from the start (slice by slice, the
    // second slice would begin one host enqueue pass -- 62 launches -- after the first)
    for (int it = P.itNext; it < itEnd; ++it) {
        for (int k = 0; k < P.nSl; ++k) {
            const PassPlan::Slice &q = P.sl[k];
            unsigned char *stp = ctx->d_psoStates + P.SB * (size_t)q.lo;
            Timed te; // events on the stream the kernel is launched on
            if (te.begin(ctx, q.st, P.useIter ? &ctx->evEval : &ctx->evEval2)) return -2;
            if (P.useIter)
                HIPCHK(pais_launch::pso_iter(sc, ctx->d_psoStates, ctx->d_active, P.cnt + 2, q.lo, q.hi, P.Nmax, P.Kmax, P.d_out,
                                             ctx->d_stat, it, 0, q.parts, ctx->d_evalBlocks, ctx->d_win, q.st));
            else if (P.useTile) {
                // many cameras: footprints staged in LDS (pais_tile.hpp); the particles it flags take the checked walk
                // (each slice's launches have their own region of the homography scratch: the slices run at the same time)
                HIPCHK(pais_launch::pso_tile(sc, stp, q.hi - q.lo, P.Nmax, P.Kmax, ctx->d_evalBlocks + P.EB * (size_t)q.lo,
                                             ctx->d_win + P.WB * (size_t)q.lo, ctx->tileStrip2, ctx->tileStrip1, ctx->tileForceNs1, ctx->tileSplit,
                                             ctx->tileStripSplit, ctx->tileBias,
                                             ctx->tileDebug ? ctx->d_stat + 8 : nullptr,
                                             (double *)((unsigned char *)ctx->d_tileH + ctx->tileHSlice * (size_t)k), ctx->tileHSlice, q.st));
                HIPCHK(pais_launch::pso_eval(sc, stp, q.hi - q.lo, P.Nmax, P.Kmax, ctx->d_evalBlocks + P.EB * (size_t)q.lo,
                                             ctx->d_win + P.WB * (size_t)q.lo, ctx->tileVerify ? 2 : 1, ctx->d_stat + 18, q.st));
            } else if (ctx->arithLiteral)
                HIPCHK(pais_launch::pso_eval_literal(sc, stp, q.hi - q.lo, P.Nmax, P.Kmax, ctx->d_evalBlocks + P.EB * (size_t)q.lo, q.st));
            else
                HIPCHK(pais_launch::pso_eval(sc, stp, q.hi - q.lo, P.Nmax, P.Kmax, ctx->d_evalBlocks + P.EB * (size_t)q.lo,
                                             ctx->d_win + P.WB * (size_t)q.lo, 0, nullptr, q.st,
                                             P.usePre ? (const double *)((const unsigned char *)ctx->d_pre + P.PB * (size_t)q.lo) : nullptr));
            if (te.end()) return -2;
            ctx->evalLaunches++;
            if (!P.useIter) ctx->eval2Launches++;
            if (P.useTile) ctx->tileLaunches++;
            if (!P.useIter)
                HIPCHK(pais_launch::pso_step(sc, P.d_out + q.lo, stp, q.hi - q.lo, P.Nmax, ctx->d_stat, q.st, ctx->d_evalBlocks + P.EB * (size_t)q.lo,
                                             P.usePre ? (double *)((unsigned char *)ctx->d_pre + P.PB * (size_t)q.lo) : nullptr, P.Kmax));
        }
    }
    if (itEnd > P.itNext) P.itNext = itEnd;
    return 0;
}

// the sub-streams a pass forked to join the context's stream again (also on the error paths: nothing stays forked)
static int pass_join(pais_ctx *ctx, PassPlan &P)
{
    for (int sI = 1; sI < P.S; ++sI) {
        bool used = false;
        for (int k = 0; k < P.nSl; ++k) used = used || (P.sl[k].st == ctx->sub[sI - 1]);
        if (!used) continue;
        HIPCHK(hipEventRecord(ctx->subDone[sI - 1], ctx->sub[sI - 1]));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->subDone[sI - 1], 0));
    }
    P.S = 1;
    return 0;
}

static int pass_close(pais_ctx *ctx, PassPlan &P)
{
    const DevScene &sc = ctx->sc;
    for (int k = 0; k < P.nSl; ++k) {
        const PassPlan::Slice &q = P.sl[k];
        // the launch after the last possible iteration only ends the runs still active
        if (P.useIter)
            HIPCHK(pais_launch::pso_iter(sc, ctx->d_psoStates, ctx->d_active, P.cnt + 2, q.lo, q.hi, P.Nmax, P.Kmax, P.d_out,
                                         ctx->d_stat, P.maxIt + 1, 1, q.parts, ctx->d_evalBlocks, ctx->d_win, q.st));
    }
    if (pass_join(ctx, P)) return -2;
    if (P.tp.end()) return -2;
    ctx->psoLaunches++;
    Timed ta;
    if (ta.begin(ctx, ctx->stream, &ctx->evAfter)) return -2;
    HIPCHK(pais_launch::after(sc, P.d_out, P.n, ctx->d_hp, P.afterGrid, P.cnt, ctx->d_stat, P.Kmax, ctx->d_ratios, P.nextCnt, ctx->stream));
    if (ta.end()) return -2;
    return 0;
}

// work buffers, the head of refine() and the set-up of every candidate's first PSO run (k_begin)
static int batch_setup(pais_ctx *ctx, PassPlan &P, int n, const pais_candidate *d_cands, pais_patch_result *d_out, int max_num_cam, int has_seeds)
{
    HIPCHK(hipSetDevice(ctx->device));
    const DevScene &sc = ctx->sc;
    int Kmax = max_num_cam > 0 ? max_num_cam : sc.numCams;
    if (Kmax > PAIS_MAX_VIS) Kmax = PAIS_MAX_VIS;
    if (Kmax > sc.numCams) Kmax = sc.numCams;
    if (Kmax < 1) Kmax = 1;
    const int Nmax = has_seeds ? sc.cfg.particleNum * 2 : sc.cfg.particleNum;
    const int S2 = sc.cfg.patchSize * sc.cfg.patchSize;
    const int afterGrid = n < 2048 ? n : 2048;
    {
        size_t hpNeed = (size_t)afterGrid * Kmax * S2 * sizeof(double);
        const size_t full = (size_t)2048 * Kmax * S2 * sizeof(double);
        if (hpNeed > ctx->hpBytes && full < ((size_t)1 << 31)) hpNeed = full;
        if (grow(ctx, ctx->d_hp, ctx->hpBytes, hpNeed)) return -2;
    }
    {
        size_t cap = ctx->activeCap * sizeof(int);
        if (grow(ctx, ctx->d_active, cap, sizeof(int) * (size_t)n)) return -2;
        ctx->activeCap = cap / sizeof(int);
    }
    if (grow(ctx, ctx->d_evalBlocks, ctx->evalBlockCap, pais_launch::eval_block_bytes_host(Kmax) * (size_t)n)) return -2;
    if (grow(ctx, ctx->d_win, ctx->winCap, pais_launch::win_bytes_per_candidate(sc) * (size_t)n)) return -2;
    const size_t SB = pais_launch::pso_state_bytes_host(Nmax);
    if (grow(ctx, ctx->d_psoStates, ctx->psoStateBytes, SB * (size_t)n)) return -2;
    if (grow(ctx, ctx->d_ratios, ctx->ratioBytes, sizeof(double) * PAIS_MAX_VIS * (size_t)n)) return -2;
    P.n = n;
    P.hasSeeds = has_seeds != 0;
    P.Nmax = Nmax;
    P.Kmax = Kmax;
    P.afterGrid = afterGrid;
    P.maxIt = has_seeds ? sc.cfg.maxIteration * 2 : sc.cfg.maxIteration;
    P.SB = SB;
    P.EB = pais_launch::eval_block_bytes_host(Kmax);
    P.WB = pais_launch::win_bytes_per_candidate(sc);
    P.d_out = d_out;
    P.d_cands = d_cands;

    if (ctx->countersDirty) HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(int) * 8, ctx->stream));
    ctx->countersDirty = true; // until this batch has run through
    Timed tb;
    if (tb.begin(ctx, ctx->stream, &ctx->evBegin)) return -2;
    // head of refine() and the set-up of the first PSO run of every candidate in one launch
    HIPCHK(pais_launch::begin(sc, d_cands, d_out, n, ctx->d_psoStates, Nmax, ctx->d_active, ctx->d_counters + 4 * (ctx->passSerial & 1) + 2,
                              ctx->d_evalBlocks, ctx->d_win, Kmax, ctx->stream));
    if (tb.end()) return -2;
    return 0;
}

// The error words of the ring pass that has just been synchronised with (h_ringCtl: per ring {head, tail, done, total, error}).
// 0: every run ended; 1: a wave ran out of patience, or runs never ended -- the caller re-runs the batch without the ring.
static int ring_failed(pais_ctx *ctx, int n)
{
    if (!ctx->ringUsed) return 0;
    ctx->ringUsed = false;
    unsigned done = 0, err = 0;
    for (int r = 0; r < PAIS_RINGS; ++r) { done += ctx->h_ringCtl[(PAIS_RING_CTL_BYTES / 4) * r + PAIS_RING_CTL_DONE_WORD]; err |= ctx->h_ringCtl[(PAIS_RING_CTL_BYTES / 4) * r + PAIS_RING_CTL_ERROR_WORD]; }
    if (err == 0 && done == (unsigned)n) return 0;
    ctx->ringFallbacks++;
    // one note per context the first time it happens in production (timeout > 0), every time with PAIS_RING_VERBOSE
    if (getenv("PAIS_RING_VERBOSE") || (ctx->ringFallbacks == 1 && ctx->ringTimeoutMs > 0))
        fprintf(stderr, "[pais] k_pso_ring did not complete (error %u, %u of %d runs ended): the batch is re-run with one launch per iteration\n", err, done, n);
    return 1;
}

// one batch on device pointers, start to end; returns 1 if a ring pass failed (the caller runs it again with the ring off)
// defer: an expansion batch whose pass ran as k_pso_ring returns without waiting for it -- the error words are looked at by
// the wire header the caller asks for next (pais_wire_header_device) and by pais_ctx_batch_status after the caller's own
// synchronisation
static int refine_device_once(pais_ctx *ctx, int n, const pais_candidate *d_cands, pais_patch_result *d_out, int max_num_cam, int has_seeds,
                              bool defer = false)
{
    PassPlan &P = ctx->plan;
    P.hostBatch = false;
    ctx->ringUsed = false;
    ctx->ringPendingN = 0;
    int rc = batch_setup(ctx, P, n, d_cands, d_out, max_num_cam, has_seeds);
    int againCount = 0; // seeds that lost cameras in the previous pass and run another PSO (patch.cpp:140-175)
    const int maxPass = has_seeds ? (PAIS_MAX_VIS + 2) : 1;
    for (int pass = 0; !rc && pass < maxPass; ++pass) {
        if ((rc = pass_open(ctx, P, pass, againCount)) != 0) break;
        if ((rc = pass_iterations(ctx, P, P.maxIt + 1)) != 0) { (void)pass_join(ctx, P); break; }
        if ((rc = pass_close(ctx, P)) != 0) break;
        if (!has_seeds && !ctx->ringUsed) break; // (asynchronous at return)
        if (!has_seeds && defer) { ctx->ringPendingN = n; break; } // (asynchronous too: the ring's status is read later)
        if (has_seeds) HIPCHK(hipMemcpyAsync(ctx->h_counters, P.cnt, sizeof(int) * 2, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ring_failed(ctx, n)) { ctx->roundHint = 0; return 1; }
        if (!has_seeds || ctx->h_counters[1] == 0) break;
        againCount = ctx->h_counters[1];
    }
    ctx->roundHint = 0;
    if (rc) return rc;
    ctx->countersDirty = false;
    return 0;
}

extern "C" int pais_refine_batch_device(pais_ctx *ctx, int n, const pais_candidate *d_cands, pais_patch_result *d_out,
                                        int max_num_cam, int has_seeds)
{
    if (!ctx || n < 0) return fail_msg("pais_refine_batch_device: bad argument");
    if (n == 0) return 0;
    if (!d_cands || !d_out) return fail_msg("pais_refine_batch_device: null pointer");
    if (ctx->openBatch >= 0) return fail_msg("pais_refine_batch_device: a batch is open on this context");
    const int hint = ctx->roundHint;
    int rc = refine_device_once(ctx, n, d_cands, d_out, max_num_cam, has_seeds);
    if (rc == 1) { // a ring pass did not complete: the same batch from its candidates, one launch per iteration
        ctx->ringSuppressed = true;
        ctx->roundHint = hint;
        rc = refine_device_once(ctx, n, d_cands, d_out, max_num_cam, has_seeds);
        ctx->ringSuppressed = false;
    }
    return rc;
}

// pais_refine_batch_device that never waits for an expansion batch (a batch with seeds is refined before it returns, as ever)
extern "C" int pais_refine_batch_device_async(pais_ctx *ctx, int n, const pais_candidate *d_cands, pais_patch_result *d_out,
                                              int max_num_cam, int has_seeds)
{
    if (!ctx || n < 0) return fail_msg("pais_refine_batch_device_async: bad argument");
    if (n == 0) return 0;
    if (!d_cands || !d_out) return fail_msg("pais_refine_batch_device_async: null pointer");
    if (ctx->openBatch >= 0) return fail_msg("pais_refine_batch_device_async: a batch is open on this context");
    const int hint = ctx->roundHint;
    const bool suppress = ctx->ringSuppressOnce;
    ctx->ringSuppressOnce = false;
    if (suppress) ctx->ringSuppressed = true;
    int rc = refine_device_once(ctx, n, d_cands, d_out, max_num_cam, has_seeds, true);
    if (rc == 1) { // (seeds: a pass's ring did not complete)
        ctx->ringSuppressed = true;
        ctx->roundHint = hint;
        rc = refine_device_once(ctx, n, d_cands, d_out, max_num_cam, has_seeds, true);
    }
    ctx->ringSuppressed = false;
    return rc;
}

// After the caller has synchronised with the context's stream: 0 = the last pais_refine_batch_device_async batch is complete;
// 1 = its k_pso_ring pass did not complete -- the records are NOT valid, the caller refines the same batch again (the next
// batch of this context then takes the per-iteration launches).
extern "C" int pais_ctx_batch_status(pais_ctx *ctx)
{
    if (!ctx) return fail_msg("pais_ctx_batch_status: bad argument");
    if (ctx->ringPendingN <= 0) return 0;
    const int n = ctx->ringPendingN;
    ctx->ringPendingN = 0;
    ctx->countersDirty = false;
    if (!ring_failed(ctx, n)) return 0;
    ctx->countersDirty = true;
    ctx->ringSuppressOnce = true;
    return 1;
}

// 64-byte header of a rank's block in the exchange of a sharded batch, written ON THE DEVICE behind the batch's launches:
// {magic 'PAIS', rc, count, rank, user_word, 0...}; rc = host_rc if that is non-zero, else PAIS_WIRE_RC_RING_RETRY when the batch's
// k_pso_ring pass did not complete (read from the ring's own words: no host round trip before the exchange), else 0.
extern "C" int pais_wire_header_device(pais_ctx *ctx, int rank, int count, int host_rc, uint32_t user_word, void *d_header)
{
    if (!ctx || !d_header) return fail_msg("pais_wire_header_device: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(pais_launch::wire_header(d_header, ctx->ringPendingN > 0 ? ctx->d_ringCtl : nullptr, ctx->ringPendingN, rank, count, host_rc, user_word, ctx->stream));
    return 0;
}

extern "C" int pais_ctx_set_round_hint(pais_ctx *ctx, int n_round)
{
    if (!ctx) return fail_msg("pais_ctx_set_round_hint: bad argument");
    ctx->roundHint = n_round > 0 ? n_round : 0;
    return 0;
}

// pais_refine_batch without the last copy: *view points at the context's pinned staging buffer (valid until the next batch
// of this context).  The drivers in pais_mvs.hip commit straight from it (2.9 MB of records per large round).
extern "C" int pais_refine_batch_view(pais_ctx *ctx, int n, const pais_candidate *cands, const pais_patch_result **view)
{
    if (!ctx || n < 0) return fail_msg("pais_refine_batch: bad argument");
    if (n == 0) return 0;
    if (!cands || !view) return fail_msg("pais_refine_batch: null pointer");
    int rc = pais_refine_batch_begin(ctx, n, cands);
    if (rc) return rc;
    return pais_refine_batch_end(ctx, view);
}

extern "C" int pais_refine_batch_begin(pais_ctx *ctx, int n, const pais_candidate *cands)
{
    int rc = pais_refine_batch_open(ctx, n, cands, 1 << 20);
    if (rc) return rc;
    rc = pais_refine_batch_enqueue(ctx, 0);
    return rc < 0 ? rc : 0;
}

// the rest of an opened batch's launches, at most `iterations` PSO iterations of them (<= 0: all); 1: the batch is enqueued
// to its end (after-stage and the copy down included), 0: iterations remain
extern "C" int pais_refine_batch_enqueue(pais_ctx *ctx, int iterations)
{
    if (!ctx) return fail_msg("pais_refine_batch_enqueue: bad argument");
    if (ctx->openBatch < 0) return fail_msg("pais_refine_batch_enqueue: no batch is open on this context");
    if (ctx->planDone) return 1;
    HIPCHK(hipSetDevice(ctx->device));
    PassPlan &P = ctx->plan;
    const int itEnd = iterations > 0 && iterations < P.maxIt + 1 - P.itNext ? P.itNext + iterations : P.maxIt + 1;
    int rc = pass_iterations(ctx, P, itEnd);
    if (!rc && P.itNext <= P.maxIt) return 0;
    if (!rc) rc = pass_close(ctx, P);
    if (rc) { (void)pass_join(ctx, P); ctx->openBatch = -1; ctx->ringUsed = false; return rc; } // (the counters stay marked dirty: cleared before the next batch)
    ctx->countersDirty = false;
    HIPCHK(hipMemcpyAsync(ctx->h_recs, ctx->d_recs, sizeof(pais_patch_result) * (size_t)ctx->openBatch, hipMemcpyDeviceToHost, ctx->stream));
    ctx->planDone = true;
    return 1;
}

extern "C" int pais_refine_batch_end(pais_ctx *ctx, const pais_patch_result **view)
{
    if (!ctx || !view) return fail_msg("pais_refine_batch_end: bad argument");
    if (ctx->openBatch < 0) return fail_msg("pais_refine_batch_end: no batch is open on this context");
    if (!ctx->planDone) {
        const int rc = pais_refine_batch_enqueue(ctx, 0);
        if (rc < 0) return rc;
    }
    ctx->openBatch = -1;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ring_failed(ctx, ctx->plan.n)) {
        // the pass did not complete: the same batch again from its candidates (still in d_cands), one launch per iteration
        PassPlan &P = ctx->plan;
        const int n = P.n;
        ctx->ringSuppressed = true;
        ctx->countersDirty = true; // (the failed pass's after-stage is not trusted to have left the counter sets clean)
        int rc = batch_setup(ctx, P, n, ctx->d_cands, ctx->d_recs, P.Kmax, 0);
        if (!rc) rc = pass_open(ctx, P, 0, 0);
        if (!rc && (rc = pass_iterations(ctx, P, P.maxIt + 1)) != 0) (void)pass_join(ctx, P);
        if (!rc) rc = pass_close(ctx, P);
        ctx->ringSuppressed = false;
        if (rc) return rc;
        ctx->countersDirty = false;
        HIPCHK(hipMemcpyAsync(ctx->h_recs, ctx->d_recs, sizeof(pais_patch_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    *view = ctx->h_recs;
    return 0;
}

extern "C" int pais_refine_batch_open(pais_ctx *ctx, int n, const pais_candidate *cands, int iterations)
{
    if (!ctx || n <= 0 || !cands) return fail_msg("pais_refine_batch_begin: bad argument");
    if (ctx->openBatch >= 0) return fail_msg("pais_refine_batch_begin: a batch is open on this context already");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ringUsed = false;
    int Kmax = 1, hasSeeds = 0;
    for (int i = 0; i < n; ++i) {
        const pais_candidate &c = cands[i];
        if (c.num_cam < 0 || c.num_cam > PAIS_MAX_VIS) return fail_msg("pais_refine_batch: num_cam out of range");
        for (int k = 0; k < c.num_cam; ++k)
            if (c.cam_idx[k] < 0 || c.cam_idx[k] >= ctx->sc.numCams) return fail_msg("pais_refine_batch: bad cam_idx");
        if (c.num_cam > Kmax) Kmax = c.num_cam;
        if (c.type == PAIS_TYPE_SEED) hasSeeds = 1;
        else if (c.type != PAIS_TYPE_EXPAND) return fail_msg("pais_refine_batch: bad type");
    }
    if ((size_t)n > ctx->recCap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_cands); (void)hipFree(ctx->d_recs);
        ctx->d_cands = nullptr; ctx->d_recs = nullptr;
        ctx->recCap = (size_t)n + (size_t)n / 2 + 256;
        HIPCHK(hipMalloc(&ctx->d_cands, sizeof(pais_candidate) * ctx->recCap));
        HIPCHK(hipMalloc(&ctx->d_recs, sizeof(pais_patch_result) * ctx->recCap));
        // pinned staging for the two per-batch copies: a pageable hipMemcpyAsync is staged by the runtime in small
        // synchronous pieces (two copy kernels and ~0.1 ms per round, which thin rounds notice)
        (void)hipHostFree(ctx->h_cands); (void)hipHostFree(ctx->h_recs);
        ctx->h_cands = nullptr; ctx->h_recs = nullptr;
        HIPCHK(hipHostMalloc((void **)&ctx->h_cands, sizeof(pais_candidate) * ctx->recCap, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&ctx->h_recs, sizeof(pais_patch_result) * ctx->recCap, hipHostMallocDefault));
    }
    memcpy(ctx->h_cands, cands, sizeof(pais_candidate) * (size_t)n);
    HIPCHK(hipMemcpyAsync(ctx->d_cands, ctx->h_cands, sizeof(pais_candidate) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (hasSeeds) {
        // the seed loop takes a host decision per pass: refined here and now
        int rc = pais_refine_batch_device(ctx, n, ctx->d_cands, ctx->d_recs, Kmax, 1);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(ctx->h_recs, ctx->d_recs, sizeof(pais_patch_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        ctx->openBatch = n;
        ctx->planDone = true;
        return 0;
    }
    PassPlan &P = ctx->plan;
    P.hostBatch = true;
    int rc = batch_setup(ctx, P, n, ctx->d_cands, ctx->d_recs, Kmax, 0);
    if (!rc) rc = pass_open(ctx, P, 0, 0);
    ctx->roundHint = 0;
    if (!rc && (rc = pass_iterations(ctx, P, iterations > 0 ? iterations : 0)) != 0) (void)pass_join(ctx, P); // (nothing stays forked behind an error)
    if (rc) { ctx->ringUsed = false; return rc; }
    ctx->openBatch = n;
    ctx->planDone = false;
    return 0;
}

extern "C" int pais_refine_batch(pais_ctx *ctx, int n, const pais_candidate *cands, pais_patch_result *out)
{
    if (n > 0 && !out) return fail_msg("pais_refine_batch: null pointer");
    const pais_patch_result *view = nullptr;
    int rc = pais_refine_batch_view(ctx, n, cands, &view);
    if (rc || n <= 0) return rc;
    memcpy(out, view, sizeof(pais_patch_result) * (size_t)n);
    return 0;
}

// ------------------------------------------------------------- record wire --
// include/pais_hip.h "wire format of a record".  Word offsets of the record are asserted against the struct.
static_assert(offsetof(pais_patch_result, imgPoint) == 136 && offsetof(pais_patch_result, key) == 1160 &&
              offsetof(pais_patch_result, type) == 1168 && offsetof(pais_patch_result, cam_idx) == 1200 &&
              offsetof(pais_patch_result, stage) == 1456 && sizeof(pais_patch_result) == 1488, "pais_patch_result layout");
static inline int wire_kw(int max_num_cam)
{
    int k = max_num_cam < 1 ? 1 : (max_num_cam > PAIS_MAX_VIS ? PAIS_MAX_VIS : max_num_cam);
    return (k + 1) & ~1;
}
extern "C" size_t pais_record_wire_bytes(int max_num_cam) { return 208 + 20 * (size_t)wire_kw(max_num_cam); }
extern "C" int pais_pack_records(int n, const pais_patch_result *recs, int max_num_cam, void *wire)
{
    if (n < 0 || (n && (!recs || !wire))) return fail_msg("pais_pack_records: bad argument");
    const int Kw = wire_kw(max_num_cam);
    const size_t WB = pais_record_wire_bytes(max_num_cam);
    for (int i = 0; i < n; ++i) {
        const unsigned char *r = (const unsigned char *)&recs[i];
        unsigned char *w = (unsigned char *)wire + WB * (size_t)i;
        memcpy(w, r, 136);                  // center .. correlation
        memcpy(w + 136, r + 1160, 8);       // key
        memcpy(w + 144, r + 1168, 32);      // type .. pso_evals
        memcpy(w + 176, r + 1456, 32);      // stage .. ncc_tables
        memcpy(w + 208, r + 136, 16 * (size_t)Kw);
        memcpy(w + 208 + 16 * (size_t)Kw, r + 1200, 4 * (size_t)Kw);
    }
    return 0;
}
extern "C" int pais_unpack_records(int n, const void *wire, int max_num_cam, pais_patch_result *recs)
{
    if (n < 0 || (n && (!recs || !wire))) return fail_msg("pais_unpack_records: bad argument");
    const int Kw = wire_kw(max_num_cam);
    const size_t WB = pais_record_wire_bytes(max_num_cam);
    for (int i = 0; i < n; ++i) {
        unsigned char *r = (unsigned char *)&recs[i];
        const unsigned char *w = (const unsigned char *)wire + WB * (size_t)i;
        memset(r, 0, sizeof(pais_patch_result));
        memcpy(r, w, 136);
        memcpy(r + 1160, w + 136, 8);
        memcpy(r + 1168, w + 144, 32);
        memcpy(r + 1456, w + 176, 32);
        memcpy(r + 136, w + 208, 16 * (size_t)Kw);
        memcpy(r + 1200, w + 208 + 16 * (size_t)Kw, 4 * (size_t)Kw);
    }
    return 0;
}
extern "C" int pais_pack_records_device(pais_ctx *ctx, int n, const pais_patch_result *d_recs, int max_num_cam, void *d_wire)
{
    if (!ctx || n < 0 || (n && (!d_recs || !d_wire))) return fail_msg("pais_pack_records_device: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(pais_launch::pack_records(d_recs, n, wire_kw(max_num_cam), d_wire, ctx->stream));
    return 0;
}

// the launches of one context (a lane is folded into its parent's figures by pais_get_kernel_stats)
static int collect_stats(pais_ctx *ctx, unsigned long long *st)
{
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (drain_events(ctx, ctx->evPso, ctx->psoMs)) return -2;
    if (drain_events(ctx, ctx->evBegin, ctx->beginMs)) return -2;
    if (drain_events(ctx, ctx->evAfter, ctx->afterMs)) return -2;
    if (drain_events(ctx, ctx->evEval, ctx->evalMs)) return -2;
    {
        double ms2 = 0;
        hipEvent_t ref = ctx->parent ? ctx->parent->refEv : ctx->refEv;
        if (ref)
            for (auto &p : ctx->evEval2) {
                float a = 0, b = 0;
                if (hipEventElapsedTime(&a, ref, p.a) == hipSuccess && hipEventElapsedTime(&b, ref, p.b) == hipSuccess)
                    ctx->eval2Intervals.push_back(std::make_pair(a, b));
            }
        if (drain_events(ctx, ctx->evEval2, ms2)) return -2;
        ctx->eval2Ms += ms2;
        ctx->evalMs += ms2;
        // the k_pso_ring launches: part of the large-batch figures above, and reported on their own
        double msR = 0;
        if (ref)
            for (auto &p : ctx->evRing) {
                float a = 0, b = 0;
                if (hipEventElapsedTime(&a, ref, p.a) == hipSuccess && hipEventElapsedTime(&b, ref, p.b) == hipSuccess)
                    ctx->eval2Intervals.push_back(std::make_pair(a, b));
            }
        if (drain_events(ctx, ctx->evRing, msR)) return -2;
        ctx->ringMs += msR;
        ctx->eval2Ms += msR;
        ctx->evalMs += msR;
    }
    HIPCHK(hipMemcpy(st, ctx->d_stat, sizeof(unsigned long long) * 24, hipMemcpyDeviceToHost));
    if (ctx->tileDebug)
        fprintf(stderr, "[pais tile] particles through the tiles %llu, DBL_MAX %llu, pending (checked walk) %llu; tiles staged %llu (%.1f KB each), cameras left in global memory %llu\n",
                st[8], st[9], st[10], st[11], st[11] ? (double)st[13] / (double)st[11] / 1024.0 : 0.0, st[12]);
    if (ctx->tileVerify) {
        double a, b;
        memcpy(&a, &st[20], 8); memcpy(&b, &st[21], 8);
        if (st[18]) // (the tests look for this line)
            fprintf(stdout, "[pais tile verify] %llu particles differ; first: candidate %llu particle %llu: tile %.17g one-wave %.17g\n", st[18],
                    st[19] >> 32, st[19] & 0xffffffffULL, a, b);
        fflush(stdout);
    }
    if (ctx->tileDebug)
        fprintf(stderr, "[pais tile] cycles of wave 0 per phase: boxes %.3g, layout %.3g, copy %.3g, walk %.3g\n", (double)st[14], (double)st[15], (double)st[16], (double)st[17]);
    return 0;
}

extern "C" int pais_get_kernel_stats(pais_ctx *ctx, pais_kernel_stats *out, int reset)
{
    if (!ctx || !out) return fail_msg("pais_get_kernel_stats: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<pais_ctx *> all(1, ctx);
    all.insert(all.end(), ctx->lanes.begin(), ctx->lanes.end());
    memset(out, 0, sizeof(*out));
    const double S2 = (double)ctx->sc.cfg.patchSize * ctx->sc.cfg.patchSize;
    std::vector<std::pair<float, float>> iv;
    for (pais_ctx *c : all) {
        unsigned long long st[24];
        const int rc = collect_stats(c, st);
        if (rc) return rc;
        out->pso_ms += c->psoMs;
        out->begin_ms += c->beginMs;
        out->after_ms += c->afterMs;
        out->pso_launches += c->psoLaunches;
        out->pso_evals += (int64_t)st[0];
        out->pso_patches += (int64_t)st[2];
        out->pso_algorithmic_bytes += (double)st[1] * S2;
        out->ncc_tables += (int64_t)st[3];
        out->ncc_algorithmic_bytes += (double)st[4] * S2 * 4.0;
        out->eval_ms += c->evalMs;
        out->eval_launches += c->evalLaunches;
        out->eval2_ms += c->eval2Ms;
        out->eval2_launches += c->eval2Launches;
        out->tile_launches += c->tileLaunches;
        out->ring_launches += c->ringLaunches;
        out->ring_ms += c->ringMs;
        out->ring_evals += (int64_t)st[22];
        out->ring_algorithmic_bytes += (double)st[23] * S2;
        out->ring_fallbacks += c->ringFallbacks;
        iv.insert(iv.end(), c->eval2Intervals.begin(), c->eval2Intervals.end());
        out->eval2_evals += (int64_t)st[5];
        out->eval2_algorithmic_bytes += (double)st[6] * S2;
        if (reset) {
            HIPCHK(hipMemset(c->d_stat, 0, sizeof(unsigned long long) * 24));
            c->psoMs = c->beginMs = c->afterMs = c->evalMs = c->eval2Ms = c->ringMs = 0;
            c->ringFallbacks = 0;
            c->psoLaunches = 0;
            c->evalLaunches = 0;
            c->eval2Launches = 0;
            c->tileLaunches = 0;
            c->ringLaunches = 0;
            c->eval2Intervals.clear();
        }
    }
    // union of the evaluation launches' intervals (all streams, all lanes)
    std::sort(iv.begin(), iv.end());
    double busy = 0;
    float curA = 0, curB = -1;
    for (auto &p : iv) {
        if (curB < curA || p.first > curB) {
            if (curB >= curA) busy += curB - curA;
            curA = p.first;
            curB = p.second;
        } else if (p.second > curB) {
            curB = p.second;
        }
    }
    if (curB >= curA) busy += curB - curA;
    out->eval2_busy_ms = busy;
    return 0;
}
