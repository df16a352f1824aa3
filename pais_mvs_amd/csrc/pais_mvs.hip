// pais_mvs.hip -- host-side reconstruction driver (no device code in this file):
// the MI355X drop-in behind MVS::refineSeedPatches() / MVS::expansionPatches()
// (TMVS/mvs/mvs.cpp:196-275).  Mirrors the reference's MVS / CellMap / Patch
// bookkeeping (mvs.cpp:529-898, cellmap.cpp) and forwards every batch of
// constructed-but-unrefined patches to pais_refine_batch().
//
// Host data structures are re-designed for 1e5..1e7 patches:
//   * cell maps: one dense int32 "head" array per camera + pooled singly linked
//     entries (the reference's vector<vector<vector<int>>> costs 24 B per empty
//     cell; the order inside a cell is irrelevant to every query the path makes)
//   * priority queue: binary heap keyed (priority, insertion sequence) with lazy
//     deletion -- the reference scans a vector<int> per pop (O(n^2), mvs.cpp:656-693);
//     ties resolve to the earliest queued id exactly as its strict '<' scan does.
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>
#include <deque>
#include <queue>
#include <new>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include "../../include/pais_mvs.h"
#include "../../include/pais_test_hooks.h"
#include "../../include/pais_seed.h"
#include "pais_dev.hpp"

#define PAIS_MAX_STREAM_PARTS 8 // parts of a streamed round on the sharded path (each on a lane of its own)
namespace {

inline int cv_round_h(double v) { return (int)lrint(v); }            // cvRound
inline int cv_ceil_h(double v) { int i = (int)v; return i + (i < v); } // cvCeil (cellmap.cpp:7-8)
inline double dot3h(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct HostCamera { // what the driver needs of PAIS::Camera (camera.h)
    double focal[2], pp[2], R[9], T[3], C[3], optN[3];
    double KR[9], KT[3]; // P = [KR | KT] (camera.cpp:123-127): the fundamental matrices of the seeding stage
    int w0, h0;
    std::vector<uint8_t> img0; // LOD-0 gray image: background test of runtimeFiltering (mvs.cpp:853-862)
};

struct HostPatch { // AbstractPatch fields the expansion loop reads (abstractpatch.h:22-53)
    pais_patch_result r;
    int id;
    int born;    // expansion round in which it was inserted (-1: seed / before expansion)
    bool expanded;
    // what every child of this patch inherits (Patch(center, parent), patch.cpp:36-43): the normal in spherical form
    // and the cameras of expandVisibleCamera (:723-761) -- functions of the parent's normal and cameras alone, built
    // when the first child is made (childCams < 0: not yet)
    bool inScene = false; // a runtimeFiltering call has kept this patch: the camera loop (mvs.cpp:851-863) passes for it
    int childCams = -1;
    double childNormalS[2];
    int childCamIdx[PAIS_MAX_VIS];
};

// what skipNeighborCell reads of a patch, one cache line per patch in a dense array (a HostPatch is a 1.5 KB heap object;
// the skip test runs for every neighbour cell of every active parent in every round)
struct HotPatch {
    double center[3], normal[3], correlation;
    int born, alive;
};

struct CellEntry { int id, next; };

// cellmap.h:15-32.  The reference holds one vector per cell (128 cameras x 2048 x 1536 cells at configs[4]: 400 M vectors).
// Here a cell is the head of a linked list in a shared pool, and the heads live in 32 x 32 tiles that are allocated when
// the first patch lands in them: setCellMaps touches a tile table (3 K entries per 12.6 MP camera), not 1.6 GB of heads.
class CellMap {
public:
    static constexpr int kTile = 32, kShift = 5, kTileCells = kTile * kTile;
    // a cell: the head of its list in the shared pool, and (round 4) what makes MVS::skipNeighborCell's answer "skip" without a
    // look at the list: the earliest expansion round in which a live patch of correlation > minCorrelation landed in the cell
    // (kNoBlock: none).  17.5 M of those tests run per reconstruction of the ring scene and seven in eight hit a cell that has
    // held such a patch for rounds -- the answer then costs the tile row alone, not the pool entry and the patch behind it
    // (two more dependent cache misses into maps far larger than the caches).
    static constexpr int32_t kNoBlock = INT32_MAX;
    struct Cell { int32_t head, block, count; }; // count: entries of the list (runtimeFiltering's full-cell test, mvs.cpp:872-886, without the walk)
    int width = 0, height = 0, tilesX = 0;
    std::vector<int32_t> tileOf; // tile table: index into `cells` / kTileCells, or -1
    std::vector<Cell> cells;
    std::vector<int32_t> claimed; // per cell: the expansion round in which a unit claimed it (R(B), round_begin); -1 never
    void init(int w, int h)
    {
        width = w; height = h;
        tilesX = (w + kTile - 1) >> kShift;
        tileOf.assign((size_t)tilesX * ((h + kTile - 1) >> kShift), -1);
        cells.clear();
        claimed.clear();
    }
    bool inMap(int x, int y) const { return !(x < 0 || y < 0 || x >= width || y >= height); } // cellmap.cpp:18-23
    bool tileEmpty(int x, int y) const { return tileOf[(size_t)(y >> kShift) * tilesX + (x >> kShift)] < 0; }
    // the cell, or nullptr while its tile does not exist
    const Cell *cell(int x, int y) const
    {
        const int32_t t = tileOf[(size_t)(y >> kShift) * tilesX + (x >> kShift)];
        return t < 0 ? nullptr : &cells[(size_t)t * kTileCells + (size_t)((y & (kTile - 1)) << kShift) + (x & (kTile - 1))];
    }
    // first pool entry of the cell, -1 if it is empty
    int32_t first(int x, int y) const
    {
        const Cell *c = cell(x, y);
        return c ? c->head : -1;
    }
    void prefetch(int x, int y) const
    {
        const Cell *c = cell(x, y);
        if (c) __builtin_prefetch(c);
    }
    // index of the cell in `cells` / `claimed`; its tile is allocated if need be (references into `cells` are valid until the
    // next call that allocates a tile)
    size_t index(int x, int y)
    {
        int32_t &t = tileOf[(size_t)(y >> kShift) * tilesX + (x >> kShift)];
        if (t < 0) {
            t = (int32_t)(cells.size() / kTileCells);
            cells.resize(cells.size() + kTileCells, Cell{-1, kNoBlock, 0});
            claimed.resize(claimed.size() + kTileCells, -1);
        }
        return (size_t)t * kTileCells + (size_t)((y & (kTile - 1)) << kShift) + (x & (kTile - 1));
    }
    // the cell's head for writing
    int32_t *slot(int x, int y) { return &cells[index(x, y)].head; }
    // first call for this cell in `round`: true (and the cell is marked); later calls of the round: false
    bool claim(int x, int y, int round)
    {
        const size_t i = index(x, y);
        if (claimed[i] == round) return false;
        claimed[i] = round;
        return true;
    }
};

struct Unit { int id, slot, j; };        // expansion attempt: neighbour j of camera slot `slot` of parent `id`
struct Candidate { // a unit that claimed its target cell this round and was sent to the GPU
    Unit u;
    int cam, cx, cy;
};
struct Active { int id, slot; }; // a popped parent and its camera-slot cursor

struct QItem {
    double pri;
    uint64_t seq;
    int id;
};
struct BestFirst { bool operator()(const QItem &a, const QItem &b) const { return a.pri > b.pri || (a.pri == b.pri && a.seq > b.seq); } };
struct WorstFirst { bool operator()(const QItem &a, const QItem &b) const { return a.pri < b.pri || (a.pri == b.pri && a.seq > b.seq); } };

} // namespace

// ------------------------------------------------------------------- RCCL ---
// librccl is loaded on first use (a single-GPU user never needs it); in a process that already holds a copy under the
// same soname (PyTorch bundles one) the loader hands that copy back.
namespace rccl {
typedef struct { char internal[PAIS_UNIQUE_ID_BYTES]; } UniqueId; // == ncclUniqueId
typedef void *Comm;                                              // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllGatherFn)(const void *, void *, size_t, int /*ncclDataType_t*/, Comm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);
struct Api {
    void *lib = nullptr;
    GetUniqueIdFn getUniqueId = nullptr;
    CommInitRankFn commInitRank = nullptr;
    CommDestroyFn commDestroy = nullptr;
    AllGatherFn allGather = nullptr;
    GetErrorStringFn errorString = nullptr;
};
static Api *api(std::string &err)
{
    static Api a;
    if (a.lib) return &a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return nullptr; }
    a.getUniqueId = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    a.commInitRank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    a.commDestroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    a.allGather = (AllGatherFn)dlsym(h, "ncclAllGather");
    a.errorString = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!a.getUniqueId || !a.commInitRank || !a.commDestroy || !a.allGather) { err = "librccl lacks an expected symbol"; return nullptr; }
    a.lib = h;
    return &a;
}
const int kInt8 = 0; // ncclInt8 / ncclChar
} // namespace rccl

// Helper threads of the enumeration of large rounds (pais_mvs_round_begin): the caller is worker 0, the pool holds the others.
class EnumPool {
public:
    explicit EnumPool(int helpers)
    {
        for (int i = 0; i < helpers; ++i) th.emplace_back([this, i] { loop(i + 1); });
    }
    ~EnumPool()
    {
        { std::lock_guard<std::mutex> l(mu); stop = true; ++gen; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    int workers() const { return (int)th.size() + 1; }
    // fn(worker) for worker = 0 .. workers() - 1; returns when all have finished
    void run(const std::function<void(int)> &fn)
    {
        { std::lock_guard<std::mutex> l(mu); job = &fn; pending = (int)th.size(); ++gen; }
        cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> l(mu);
        cvDone.wait(l, [this] { return pending == 0; });
        job = nullptr;
    }
private:
    void loop(int worker)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)> *f;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return gen != seen; });
                seen = gen;
                if (stop) return;
                f = job;
            }
            (*f)(worker);
            { std::lock_guard<std::mutex> l(mu); if (--pending == 0) cvDone.notify_one(); }
        }
    }
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvDone;
    const std::function<void(int)> *job = nullptr;
    unsigned long gen = 0;
    int pending = 0;
    bool stop = false;
};

struct WorkItem { Unit u; int cam, x, y; };

struct pais_mvs {
    pais_config cfg;
    pais_ctx *ctx = nullptr;
    int device = -1;
    // ---- multi-GPU
    int rank = 0, world = 1;
    rccl::Comm nccl = nullptr;
    pais_allgather_fn gatherCb = nullptr;
    void *gatherUser = nullptr;
    int replicateBelow = PAIS_REPLICATE_BELOW_WAVES;
    pais_record_source_fn recordSource = nullptr;
    void *recordUser = nullptr;
    // a sharded batch in flight (two at most: the parts of a streamed round): this rank's shard on the device, its wire block,
    // every rank's blocks as the exchange delivers them
    struct ShardBufs {
        pais_candidate *d_shardC = nullptr, *h_shardC = nullptr;   // this rank's shard of a batch (device / pinned)
        pais_patch_result *d_shardR = nullptr;
        size_t shardCap = 0;
        void *d_wireS = nullptr, *d_wireAll = nullptr;             // wire slots: this rank's (header + shard), all ranks'
        unsigned char *h_wireAll = nullptr;                        // pinned
        size_t wireCap = 0;   // bytes of d_wireAll / h_wireAll (every rank's block)
        size_t wireSCap = 0;  // bytes of d_wireS (this rank's block): tracked on its own -- a smaller world makes the slot larger
        hipEvent_t packed = nullptr, done = nullptr;
        // the same buffers of the HOST instance of the protocol (drivers without RCCL: a caller-supplied all-gather, GPU-less
        // schedulers under test) -- shard_submit / shard_finish run the same header / growth / retry protocol over them
        std::vector<pais_candidate> hostC;
        std::vector<pais_patch_result> hostR;
        std::vector<unsigned char> hostS, hostAll;
    } sb[PAIS_MAX_STREAM_PARTS];
    struct ShardXfer {
        ShardBufs *B = nullptr;
        pais_ctx *lane = nullptr;
        const pais_candidate *c = nullptr;
        int n = 0, per = 0, lo = 0, cnt = 0, Kb = 1, Kmax = 1, hasSeeds = 0, localRc = 0;
        size_t WB = 0, slot = 0;
        double t0 = 0;
        bool sharded = false, open = false;                        // sharded: device path + exchange; else replicated on `lane` (host batch)
        bool hostRetry = false;                                    // host instance: this rank's block said "ring retry" (test hook)
        bool secondAttempt = false;                                // the shard is being refined again after such a status
    } sx[PAIS_MAX_STREAM_PARTS];
    // pais_mvs_test_inject (include/pais_test_hooks.h): failures of the sharded protocol provoked on this rank
    int injectRingRetry = 0, injectGrowthFail = 0, injectRefineFail = 0;
    int *d_hs = nullptr, *d_hsAll = nullptr, *h_hs = nullptr;      // growth handshake of the sharded path (4 bytes per rank)
    // one-GPU emulation of a rank of a larger world (pais_mvs_emulate): the sharded code path with the other ranks' blocks
    // replayed from the records of a single-rank run of the same workload
    int emuMode = 0;                                               // 0 off, 1 record (single rank), 2 emulate
    std::unordered_map<uint64_t, uint32_t> emuIndex;               // candidate key -> position in emuRecs
    std::vector<pais_patch_result> emuRecs;
    double emuLatencyUs = 25.0;                                    // PAIS_EMU_LATENCY_US: modelled launch latency of the collective
    std::vector<HostCamera> cams;
    std::vector<HostPatch *> patches; // index == id; nullptr once deleted  (map<int,Patch>, mvs.h:86)
    int alive = 0;
    bool deepPrefetch = false;         // round_begin: also prefetch pool entries / patches (maps far beyond the caches)
    bool trustSceneStage = true;       // PAIS_HOST_SCENE_TEST=1: always evaluate the camera loop of runtimeFiltering on the host
    std::vector<HotPatch> hot;         // hot[id]: dense copy of what skipNeighborCell reads of patches[id]
    std::vector<CellMap> cellMaps;     // mvs.h:88 (empty until setCellMaps)
    bool blockEnable = true;           // PAIS_CELL_BLOCK=0: skipNeighborCell always walks the lists (A/B, tests)
    bool blocksValid = false;          // the cells' `block` summaries describe the lists (false while a patch's correlation has changed
                                       // under existing maps: skipNeighborCell then walks the lists as the reference does)
    std::vector<CellEntry> pool;
    int freeEntry = -1;
    std::priority_queue<QItem, std::vector<QItem>, BestFirst> qBest;
    std::priority_queue<QItem, std::vector<QItem>, WorstFirst> qWorst;
    std::deque<int> qList;             // breadth / depth strategies keep the literal container
    uint64_t qSeq = 0;
    long liveQueued = 0;               // alive, unexpanded, queued patches == the reference's queue.size() after a pop
    double neighborRadius = 0;
    // round state
    std::vector<Active> active;        // ordered active set of the slot-synchronous rounds
    std::vector<Unit> deferred, nextDeferred; // cell-claim rule: units retried at the head of the next round
    int curRound = -1;
    int thinFront = PAIS_DEFAULT_THIN_FRONT; // rounds with <= thinFront active parents take all remaining slots of each parent
    bool queueExhausted = false;
    std::vector<Candidate> cands;
    std::vector<pais_candidate> candRecs;
    std::vector<pais_patch_result> results;
    std::vector<int> seedIds;
    bool strictTail = true;
    long truncatedVisible = 0;         // cameras dropped because a visibility cone held more than PAIS_MAX_VIS of them
    pais_mvs_stats st;
    std::vector<pais_round_log> roundLog;
    double lastEnumerateMs = 0;
    // enumeration of large rounds on several host threads (round_begin)
    EnumPool *enumPool = nullptr;
    int enumThreads = 0;               // PAIS_ENUM_THREADS (default min(8, hardware threads)); 1 = always the single-thread walk
    size_t enumThreadsAbove = 4096;    // PAIS_ENUM_ABOVE: units of a round from which the threads are used
    std::vector<WorkItem> work;
    std::vector<unsigned char> workState;
    std::vector<std::vector<uint32_t>> workBucket;
    // sharded enumeration (round 6, opt-in: PAIS_SHARD_ENUM=1): in a world of N ranks the skip test + claim of a
    // round's units -- the cache-miss chains that are most of a large round's listing -- are dealt to the ranks by (camera, tile)
    // exactly as the threaded walk deals them to threads; the unit states (one byte each) are exchanged with one small
    // all-gather and merged; every rank then builds the same candidate list.  Rounds handled this way are not streamed.
    int shardEnum = 0;
    size_t shardEnumAbove = 2048;      // PAIS_SHARD_ENUM_ABOVE: units of a round from which the listing is sharded
    std::vector<unsigned char> enumAll; // host: world x units
    unsigned char *d_en = nullptr, *d_enAll = nullptr, *h_en = nullptr; // RCCL instance of the exchange (device + pinned host)
    size_t enCap = 0;
    // streamed rounds (pais_mvs_expansion_patches on one GPU): the candidates of the first part of a round's work list are on
    // the GPU (this context) while the host enumerates the rest (-> lane1), and the first part's records are committed while
    // the second part is still being refined.  Same work list, same commit order: same result.
    pais_ctx *lane1 = nullptr;         // owned by ctx (pais_ctx_fork_lane)
    int streamRounds = 1;              // PAIS_STREAM_ROUNDS: 0 every round is one batch; 1 rounds whose host work is worth hiding (below); 2 every
                                       // round of at least streamAbove active parents
    // Streaming costs about 0.3 ms of GPU time per round (each part runs alone for a while, with one sub-stream) and hides at most
    // the enumeration of the second part and the commit of the first: measured +5 % on the ring (1.7 ms of host work per round
    // of 17 ms), -2 % on the pawn scene (0.45 of 6 ms).  A round is streamed when the previous round's host work was at
    // least streamHostMs and at least streamHostShare of its GPU time -- a choice that never changes a record.
    double streamHostMs = 0.9, streamHostShare = 0.04; // PAIS_STREAM_HOST_MS / PAIS_STREAM_HOST_SHARE
    double prevHostMs = 0, prevGpuMs = 0;
    size_t streamAbove = 192;          // PAIS_STREAM_ABOVE: active parents from which a round is streamed
    double streamSplit = 0.5;          // PAIS_STREAM_SPLIT: share of the active parents in the first part
    // Whether a round is streamed is a TIMING-driven choice (below), and on the sharded path every rank must take the same one
    // (a streamed round is several collectives, an unstreamed one a single larger one): each rank writes its own wish for the
    // next round into the status header of its block (streamWish), and rank 0's word, which every rank reads, decides
    // (streamAgreed) -- no collective of its own.
    bool streamWish = false, streamAgreed = false;
    // rounds of the multi-GPU path streamed in parts (several collectives in flight per round): 1 on, 0 off,
    // -1 (default) = on in the one-GPU emulation, where it has been measured, OFF over a real RCCL communicator until a run on
    // >= 2 real GPUs has been green (that path has only ever run against the emulated transport)
    int streamSharded = -1;
    // streamed rounds over a REAL communicator (streamSharded < 0, the default): off until one streamed sharded round -- the canary --
    // has been refined a second time as ONE unstreamed sharded batch and every rank has found the two record sets identical, byte
    // for byte (agreed through the 4-byte handshake); 0 not tried yet, 1 verified: stream, -1 differed on some rank: never stream.
    // PAIS_STREAM_CANARY=1 applies the same rule to the one-GPU emulation (tests), =0 skips the canary (stream unverified: round 5)
    int streamCanary = 0;
    int streamCanaryMode = -1;
    std::vector<pais_patch_result> canaryRecs;
    // how a batch is dealt to the ranks: false (default) = contiguous count-balanced shards, true = candidate i to rank i mod world
    // (PAIS_SHARD_STRIDED=1; must be the same on every rank).  Round 5 built the second rule against a suspected cost gradient along
    // the work list and then measured EVERY rank of the emulated 8-rank dome: contiguous shards are balanced already (1.40 ... 1.48 s
    // over the eight ranks), dealing the candidates out is no better (profiles/r05_emu_dome40_by_rank.txt) -- the default stays.
    bool shardStrided = false;
    int streamHead = 4, streamStep = 2; // PAIS_STREAM_HEAD / _STEP: PSO iterations of a part enqueued when it is opened / per turn after that
    const std::function<int(const pais_candidate *, int)> *onFirstPart = nullptr; // set by the streamed driver for one round_begin
    int firstPart = -1;                // candidates of the first part of the current round (-1: the round is one batch)
    // the sharded path streams a round in up to streamParts parts (round 4): with N GPUs the GPU's share of a round shrinks N-fold
    // and the replicated host work does not, so more of it has to run under the GPU's -- part p is on the GPU while the host lists
    // part p + 1, and is committed while the later parts are still being refined; what stays exposed is the listing of the first
    // part and the commit of the last
    int streamParts = 4;               // PAIS_STREAM_PARTS
    double streamFirst = 0.0;          // PAIS_STREAM_FIRST: share of the round's parents in the first part (0: an equal share)
    const std::function<int(int, const pais_candidate *, int)> *onPart = nullptr; // (part index, its candidates, their number)
    std::vector<int> partEnd;          // candidates listed when part p was handed out (parts 0 .. size-1; the last part is the rest)
    std::vector<pais_ctx *> partLanes; // lanes of parts 1 .. (owned by ctx)
    std::string err;

    ~pais_mvs()
    {
        delete enumPool;
        for (void *c : patchChunks) free(c);
        if (device >= 0) {
            (void)hipSetDevice(device);
            if (nccl) {
                std::string e;
                if (rccl::Api *a = rccl::api(e)) a->commDestroy(nccl);
            }
            for (ShardBufs &b : sb) {
                (void)hipFree(b.d_shardC); (void)hipFree(b.d_shardR); (void)hipFree(b.d_wireS); (void)hipFree(b.d_wireAll);
                (void)hipHostFree(b.h_shardC); (void)hipHostFree(b.h_wireAll);
                if (b.packed) (void)hipEventDestroy(b.packed);
                if (b.done) (void)hipEventDestroy(b.done);
            }
            (void)hipFree(d_hs); (void)hipFree(d_hsAll); (void)hipHostFree(h_hs);
            (void)hipFree(d_en); (void)hipFree(d_enAll); (void)hipHostFree(h_en);
        }
        if (ctx) pais_ctx_destroy(ctx);
    }

    // ---- Camera::project / inImage (camera.cpp:138-160, camera.h:116-131), LOD 0
    bool project0(int cam, const double *X, double *out) const
    {
        const HostCamera &c = cams[cam];
        pais::project_raw(c.R, c.T, c.focal, c.pp, 1.0, X, out);
        return pais::in_image_d(out, c.w0, c.h0);
    }

    // ---- cell maps
    int cellCount(const CellMap &m, int x, int y) const
    {
        int n = 0;
        for (int e = m.first(x, y); e >= 0; e = pool[e].next) ++n;
        return n;
    }
    bool cellInsert(CellMap &m, int x, int y, int id) // cellmap.cpp:25-29
    {
        if (!m.inMap(x, y)) return false;
        int e;
        if (freeEntry >= 0) { e = freeEntry; freeEntry = pool[e].next; }
        else { e = (int)pool.size(); pool.push_back(CellEntry()); }
        CellMap::Cell &c = m.cells[m.index(x, y)];
        pool[e].id = id;
        pool[e].next = c.head;
        c.head = e;
        c.count++;
        const HotPatch &hp = hot[id];
        if (hp.alive && hp.correlation > cfg.minCorrelation && hp.born < c.block) c.block = hp.born;
        return true;
    }
    // the cell's summary from its list (after a patch has left it)
    void cellReblock(CellMap::Cell &c) const
    {
        int32_t b = CellMap::kNoBlock;
        for (int e = c.head; e >= 0; e = pool[e].next) {
            const HotPatch &hp = hot[pool[e].id];
            if (hp.alive && hp.correlation > cfg.minCorrelation && hp.born < b) b = hp.born;
        }
        c.block = b;
    }
    bool cellDrop(CellMap &m, int x, int y, int id) // cellmap.cpp:31-38
    {
        if (!m.inMap(x, y)) return false;
        if (m.first(x, y) < 0) return false;
        int32_t *link = m.slot(x, y);
        while (*link >= 0) {
            int e = *link;
            if (pool[e].id == id) {
                *link = pool[e].next;
                pool[e].next = freeEntry;
                freeEntry = e;
                CellMap::Cell &c = m.cells[m.index(x, y)];
                c.count--;
                cellReblock(c);
                return true;
            }
            link = &pool[e].next;
        }
        return false;
    }

    // ---- Patch::isNeighbor, patch.cpp:6-23
    bool isNeighbor(const pais_patch_result &a, const pais_patch_result &b) const
    {
        double d[3] = {a.center[0] - b.center[0], a.center[1] - b.center[1], a.center[2] - b.center[2]};
        double dist = 0;
        dist += fabs(dot3h(d, a.normal));
        dist += fabs(dot3h(d, b.normal));
        return dist <= neighborRadius;
    }

    // ---- MVS::skipNeighborCell, mvs.cpp:792-807.  beforeRound >= 0: on the state before that round
    //      (patches inserted during it are ignored; cell-claim rule of R(B))
    bool skipNeighborCell(const CellMap &m, int x, int y, const pais_patch_result &ref, int beforeRound) const
    {
        const CellMap::Cell *cl = m.cell(x, y);
        if (!cl || cl->head < 0) return false;
        // a live patch of correlation > minCorrelation that was in the cell before the round makes the second loop below return
        // true (if the count has not done so already): the cell's summary says so without the walk
        if (blocksValid && (beforeRound >= 0 ? cl->block < beforeRound : cl->block != CellMap::kNoBlock)) return true;
        const int first = cl->head;
        int pthNum = 0;
        for (int e = first; e >= 0; e = pool[e].next) {
            const HotPatch &p = hot[pool[e].id];
            if (beforeRound >= 0 && p.alive && p.born >= beforeRound) continue;
            ++pthNum;
        }
        if (pthNum >= cfg.maxCellPatchNum) return true;
        for (int e = first; e >= 0; e = pool[e].next) {
            const HotPatch &p = hot[pool[e].id];
            if (!p.alive) continue;
            if (beforeRound >= 0 && p.born >= beforeRound) continue;
            if (p.correlation > cfg.minCorrelation) return true;
            // Patch::isNeighbor (above) on the dense copy
            const double d[3] = {ref.center[0] - p.center[0], ref.center[1] - p.center[1], ref.center[2] - p.center[2]};
            double dist = 0;
            dist += fabs(dot3h(d, ref.normal));
            dist += fabs(dot3h(d, p.normal));
            if (dist <= neighborRadius) return true;
        }
        return false;
    }

    // ---- MVS::getExpansionPatchCenter, mvs.cpp:809-836
    void expansionCenter(int camI, const pais_patch_result &parent, int cx, int cy, double *center) const
    {
        const HostCamera &cam = cams[camI];
        const double px = (cx + 0.5) * cfg.cellSize;
        const double py = (cy + 0.5) * cfg.cellSize;
        double p3d[3], tmp[3];
        p3d[0] = (px - cam.pp[0]) / cam.focal[0];
        p3d[1] = (py - cam.pp[1]) / cam.focal[1];
        p3d[2] = 1.0;
        for (int i = 0; i < 3; ++i) tmp[i] = p3d[i] - cam.T[i];
        for (int i = 0; i < 3; ++i) p3d[i] = cam.R[0 * 3 + i] * tmp[0] + cam.R[1 * 3 + i] * tmp[1] + cam.R[2 * 3 + i] * tmp[2];
        double v13[3], v12[3];
        for (int i = 0; i < 3; ++i) v13[i] = parent.center[i] - cam.C[i];
        for (int i = 0; i < 3; ++i) v12[i] = p3d[i] - cam.C[i];
        const double u = dot3h(parent.normal, v13) / dot3h(parent.normal, v12);
        for (int i = 0; i < 3; ++i) center[i] = cam.C[i] + u * v12[i];
    }

    // ---- Patch(center, parent) constructor, patch.cpp:36-43 incl. expandVisibleCamera :723-761
    void makeExpandCandidate(HostPatch *hp, const double *center, uint64_t key, pais_candidate *c)
    {
        const pais_patch_result &parent = hp->r;
        memset(c, 0, sizeof(*c));
        for (int i = 0; i < 3; ++i) { c->center[i] = center[i]; c->normal[i] = parent.normal[i]; }
        c->key = key;
        c->type = PAIS_TYPE_EXPAND;
        if (hp->childCams >= 0) { // a sibling was made before: same normal, same cameras
            c->normalS[0] = hp->childNormalS[0];
            c->normalS[1] = hp->childNormalS[1];
            c->num_cam = hp->childCams;
            for (int i = 0; i < hp->childCams; ++i) c->cam_idx[i] = hp->childCamIdx[i];
            return;
        }
        pais::normal2spherical(c->normal, c->normalS); // setNormal(Vec3d), abstractpatch.cpp:43-46
        int exp[PAIS_MAX_VIS * 2];
        int n = 0;
        for (int i = 0; i < (int)cams.size(); ++i) {
            double neg[3] = {-cams[i].optN[0], -cams[i].optN[1], -cams[i].optN[2]};
            if (dot3h(c->normal, neg) >= cfg.visibleCorrelation) {
                if (n < PAIS_MAX_VIS) exp[n++] = i;
                else ++truncatedVisible; // more than PAIS_MAX_VIS cameras in the cone: the reference would keep them all
            }
        }
        if (n < cfg.minCamNum) {
            for (int i = 0; i < parent.num_cam; ++i) {
                const HostCamera &cam = cams[parent.cam_idx[i]];
                double neg[3] = {-cam.optN[0], -cam.optN[1], -cam.optN[2]};
                if (dot3h(c->normal, neg) >= cfg.visibleCorrelation / 2.0 && n < PAIS_MAX_VIS * 2) exp[n++] = parent.cam_idx[i];
            }
            std::sort(exp, exp + n);
            n = (int)(std::unique(exp, exp + n) - exp);
        }
        if (n > PAIS_MAX_VIS) n = PAIS_MAX_VIS;
        c->num_cam = n; // < minCamNum => refine() drops it (patch.cpp:118-123), as `drop` would
        for (int i = 0; i < n; ++i) c->cam_idx[i] = exp[i];
        hp->childCams = n;
        hp->childNormalS[0] = c->normalS[0];
        hp->childNormalS[1] = c->normalS[1];
        for (int i = 0; i < n; ++i) hp->childCamIdx[i] = exp[i];
    }

    // ---- MVS::runtimeFiltering, mvs.cpp:838-898
    //      scene: what is known of the camera loop (:851-863), a function of the record and the scene alone:
    //      0 nothing -- evaluated here; 1 it passes (the device evaluated it, or the patch was kept by an earlier call);
    //      2 it fails
    bool runtimeFiltering(const pais_patch_result &p, int id, int scene = 0) const
    {
        if (p.dropped) return false;
        if (p.num_cam < cfg.minCamNum) return false;
        if (p.fitness > cfg.maxFitness) return false;
        if (p.fitness == 0.0) return false;
        if (p.priority > 10000) return false;
        if (std::isnan(p.fitness)) return false;
        if (std::isnan(p.priority)) return false;
        if (std::isnan(p.correlation)) return false;
        if (p.correlation < cfg.minCorrelation) return false;
        if (scene == 2) return false;
        double pt[2];
        for (int i = 0; scene == 0 && i < (int)cams.size(); i++) {
            if (!project0(i, p.center, pt)) return false;
            const HostCamera &cam = cams[i];
            // mvs.cpp:860 reads at(cvRound(y), cvRound(x)) after 0 <= pt < dim only: within half a pixel of the right /
            // bottom edge that is x == cols / y == rows, out of bounds in the reference.  Defined as the edge pixel, so
            // that replicated drivers can never disagree on what lies behind the buffer.
            const int rx = std::min(cv_round_h(pt[0]), cam.w0 - 1), ry = std::min(cv_round_h(pt[1]), cam.h0 - 1);
            if (cam.img0[(size_t)ry * cam.w0 + rx] == 0) return false;
        }
        int count = 0;
        for (int i = 0; i < p.num_cam; ++i) {
            const HostCamera &cam = cams[p.cam_idx[i]];
            double neg[3] = {-cam.optN[0], -cam.optN[1], -cam.optN[2]};
            if (dot3h(p.normal, neg) > 0) count++;
        }
        if (count < cfg.minCamNum) return false;
        if (cellMaps.empty()) return true;
        int fullCellCounter = 0;
        for (int i = 0; i < p.num_cam; ++i) {
            int cx = (int)(p.imgPoint[i][0] / cfg.cellSize);
            int cy = (int)(p.imgPoint[i][1] / cfg.cellSize);
            const CellMap &m = cellMaps[p.cam_idx[i]];
            if (!m.inMap(cx, cy)) continue;
            const CellMap::Cell *cl = m.cell(cx, cy);
            if (!cl) continue;
            // (a patch that is being inserted -- id == the next slot -- is in no list yet: only the count matters)
            bool found = false;
            if (id < (int)patches.size())
                for (int e = cl->head; e >= 0; e = pool[e].next)
                    if (pool[e].id == id) { found = true; break; }
            if (found) return true;
            if (cl->count >= cfg.maxCellPatchNum) ++fullCellCounter;
        }
        if (fullCellCounter >= p.num_cam) return false;
        return true;
    }

    // HostPatch records come from an arena of fixed-size chunks (no malloc, no zero fill and -- from the second
    // reconstruction of a driver on -- no page faults per inserted patch: 10 k patches x 1.8 KB per pawn reconstruction)
    static constexpr size_t kPatchChunk = 1024;
    std::vector<void *> patchChunks;
    size_t patchesUsed = 0;
    HostPatch *allocPatch()
    {
        if (patchesUsed == patchChunks.size() * kPatchChunk) {
            void *c = malloc(sizeof(HostPatch) * kPatchChunk);
            if (!c) throw std::bad_alloc();
            patchChunks.push_back(c);
        }
        HostPatch *slot = (HostPatch *)patchChunks[patchesUsed / kPatchChunk] + (patchesUsed % kPatchChunk);
        ++patchesUsed;
        return new (slot) HostPatch; // default-initialised: the record is assigned by the caller
    }

    // a record into the arena: what is meaningful of it -- the scalars and the first num_cam entries of the two PAIS_MAX_VIS-long
    // arrays (1 280 of its 1 488 bytes are those arrays; a pawn patch uses 100 of them).  The array tails of an arena record
    // are unspecified; pais_mvs_get_patch hands out zeros there, which is what the batch calls write.
    static void copyRecordCompact(pais_patch_result *dst, const pais_patch_result &r)
    {
        static_assert(offsetof(pais_patch_result, imgPoint) == 136 && offsetof(pais_patch_result, key) == 1160 &&
                      offsetof(pais_patch_result, cam_idx) == 1200 && offsetof(pais_patch_result, stage) == 1456 &&
                      sizeof(pais_patch_result) == 1488, "pais_patch_result layout");
        const int K = r.num_cam < 0 ? 0 : (r.num_cam > PAIS_MAX_VIS ? PAIS_MAX_VIS : r.num_cam);
        const unsigned char *s = (const unsigned char *)&r;
        unsigned char *d = (unsigned char *)dst;
        memcpy(d, s, 136);
        memcpy(d + 136, s + 136, sizeof(double) * 2 * (size_t)K);
        memcpy(d + 1160, s + 1160, 40);
        memcpy(d + 1200, s + 1200, sizeof(int32_t) * (size_t)K);
        memcpy(d + 1456, s + 1456, 32);
    }

    int storePatch(const pais_patch_result &r)
    {
        HostPatch *hp = allocPatch();
        copyRecordCompact(&hp->r, r);
        hp->id = (int)patches.size();
        hp->expanded = false;
        hp->born = curRound;
        patches.push_back(hp);
        HotPatch h;
        for (int i = 0; i < 3; ++i) { h.center[i] = r.center[i]; h.normal[i] = r.normal[i]; }
        h.correlation = r.correlation;
        h.born = hp->born;
        h.alive = 1;
        hot.push_back(h);
        ++alive;
        return hp->id;
    }

    void queuePush(int id)
    {
        QItem q{patches[id]->r.priority, qSeq++, id};
        switch (cfg.expansionStrategy) {
        default:
        case 0: qBest.push(q); break;
        case 1: qWorst.push(q); break;
        case 2:
        case 3: qList.push_back(id); break;
        }
        ++liveQueued;
    }
    bool queueLive(int id) const { return id >= 0 && id < (int)patches.size() && patches[id] && !patches[id]->expanded; }
    // MVS::getPatchIdFromQueue (mvs.cpp:632-788)
    int queuePop()
    {
        int id = -1;
        switch (cfg.expansionStrategy) {
        default:
        case 0:
            while (!qBest.empty()) { QItem q = qBest.top(); qBest.pop(); if (queueLive(q.id)) { id = q.id; break; } }
            break;
        case 1:
            while (!qWorst.empty()) { QItem q = qWorst.top(); qWorst.pop(); if (queueLive(q.id)) { id = q.id; break; } }
            break;
        case 2: // breadth first :734-759
            while (!qList.empty()) { int q = qList.front(); qList.pop_front(); if (queueLive(q)) { id = q; break; } }
            break;
        case 3: { // depth first :761-788 -- never examines queue[0]
            while (qList.size() > 1) {
                int q = qList.back();
                if (!queueLive(q)) { qList.pop_back(); continue; }
                id = q;
                qList.pop_back();
                break;
            }
            if (id < 0 && !qList.empty()) qList.pop_front(); // queue.erase(begin) with topId == -1
            break;
        }
        }
        if (id >= 0) --liveQueued;
        return id;
    }

    // ---- MVS::deletePatch, mvs.cpp:603-630
    void deletePatch(int id)
    {
        HostPatch *p = (id >= 0 && id < (int)patches.size()) ? patches[id] : nullptr;
        if (!p) return;
        if (!cellMaps.empty()) {
            for (int i = 0; i < p->r.num_cam; ++i) {
                int cx = (int)(p->r.imgPoint[i][0] / cfg.cellSize);
                int cy = (int)(p->r.imgPoint[i][1] / cfg.cellSize);
                cellDrop(cellMaps[p->r.cam_idx[i]], cx, cy, id);
            }
        }
        patches[id] = nullptr; // (the record's memory goes back with the arena: pais_mvs_reset / destroy)
        hot[id].alive = 0;
        --alive;
        st.patches_deleted++;
    }

    // what the batch call reports of the scene half of runtimeFiltering (include/pais_hip.h PAIS_DONE_*)
    int sceneKnowledge(const pais_patch_result &r) const
    {
        if (!trustSceneStage) return 0;
        return r.stage == PAIS_DONE_IN_SCENE ? 1 : (r.stage == PAIS_DONE_OFF_SCENE ? 2 : 0);
    }

    // ---- MVS::insertPatch, mvs.cpp:579-601
    bool insertPatch(const pais_patch_result &r)
    {
        const int newId = (int)patches.size();
        if (!runtimeFiltering(r, newId, sceneKnowledge(r))) return false;
        int id = storePatch(r);
        patches[id]->inScene = true;
        queuePush(id);
        for (int i = 0; i < r.num_cam; ++i) {
            int cx = (int)(r.imgPoint[i][0] / cfg.cellSize);
            int cy = (int)(r.imgPoint[i][1] / cfg.cellSize);
            cellInsert(cellMaps[r.cam_idx[i]], cx, cy, id);
        }
        st.patches_inserted++;
        return true;
    }

    // ---- MVS::setNeighborRadius + getBoundingVolume, mvs.cpp:147-152, 974-997
    void setNeighborRadius()
    {
        double minP[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, maxP[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (auto *p : patches) {
            if (!p) continue;
            for (int i = 0; i < 3; ++i) {
                if (p->r.center[i] < minP[i]) minP[i] = p->r.center[i];
                if (p->r.center[i] > maxP[i]) maxP[i] = p->r.center[i];
            }
        }
        double vol[3] = {maxP[0] - minP[0], maxP[1] - minP[1], maxP[2] - minP[2]};
        double volume = fabs(vol[0] * vol[1] * vol[2]);
        neighborRadius = pow(volume, 1.0 / 3.0) * cfg.neighborRadiusScalar;
        cfg.neighborRadius = neighborRadius;
        if (ctx) pais_ctx_set_neighbor_radius(ctx, neighborRadius);
    }

    // ---- MVS::setCellMaps + initPriorityQueue, mvs.cpp:89-95, 116-133
    void setCellMaps()
    {
        cellMaps.assign(cams.size(), CellMap());
        pool.clear();
        freeEntry = -1;
        size_t cellBytes = 0;
        for (size_t c = 0; c < cams.size(); ++c) {
            cellMaps[c].init(cv_ceil_h((double)cams[c].w0 / (double)cfg.cellSize), cv_ceil_h((double)cams[c].h0 / (double)cfg.cellSize));
            cellBytes += sizeof(int32_t) * (size_t)cellMaps[c].width * cellMaps[c].height;
        }
        deepPrefetch = cellBytes > ((size_t)16 << 20);
        blocksValid = blockEnable;
        for (auto *p : patches) {
            if (!p) continue;
            for (int i = 0; i < p->r.num_cam; ++i) {
                int cx = (int)(p->r.imgPoint[i][0] / cfg.cellSize);
                int cy = (int)(p->r.imgPoint[i][1] / cfg.cellSize);
                cellInsert(cellMaps[p->r.cam_idx[i]], cx, cy, p->id);
            }
        }
    }
    void initPriorityQueue()
    {
        qBest = decltype(qBest)();
        qWorst = decltype(qWorst)();
        qList.clear();
        qSeq = 0;
        liveQueued = 0;
        for (auto *p : patches)
            if (p) queuePush(p->id);
    }
};

// ------------------------------------------------------------------ C ABI ---
static thread_local std::string g_mvs_err;
static int mfail(const char *m) { g_mvs_err = m; return -1; }

extern "C" int pais_mvs_create(const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device,
                               uint64_t pso_seed, pais_mvs **out)
{
    if (!cfg || !cams || !out || num_cams <= 0) return mfail("pais_mvs_create: bad argument");
    pais_mvs *m = new pais_mvs();
    if (const char *e = getenv("PAIS_HOST_SCENE_TEST")) m->trustSceneStage = atoi(e) == 0;
    if (const char *e = getenv("PAIS_CELL_BLOCK")) m->blockEnable = atoi(e) != 0;
    if (const char *e = getenv("PAIS_THIN_FRONT")) m->thinFront = atoi(e) < 0 ? 0 : atoi(e); // tuning sweeps (scripts/)
    {
        const unsigned hw = std::thread::hardware_concurrency();
        (void)hw;
        m->enumThreads = 1; // measured on the GPU box (2 x 128 cores): pawn enumerate 4.1 -> 8.6 ms, ring 1.17 -> 1.93 s with 8 threads --
                            // a round's walk is too short for the wake-ups; the threaded walk stays selectable (and tested)
        if (const char *e = getenv("PAIS_ENUM_THREADS")) m->enumThreads = std::max(1, std::min(64, atoi(e)));
        if (const char *e = getenv("PAIS_ENUM_ABOVE")) m->enumThreadsAbove = (size_t)std::max(0, atoi(e));
    }
    if (const char *e = getenv("PAIS_SHARD_ENUM")) m->shardEnum = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("PAIS_SHARD_ENUM_ABOVE")) m->shardEnumAbove = (size_t)std::max(0, atoi(e));
    if (const char *e = getenv("PAIS_STREAM_ROUNDS")) m->streamRounds = atoi(e);
    if (const char *e = getenv("PAIS_STREAM_ABOVE")) m->streamAbove = (size_t)std::max(0, atoi(e));
    if (const char *e = getenv("PAIS_STREAM_HOST_MS")) m->streamHostMs = atof(e);
    if (const char *e = getenv("PAIS_STREAM_HOST_SHARE")) m->streamHostShare = atof(e);
    if (const char *e = getenv("PAIS_STREAM_HEAD")) m->streamHead = std::max(1, atoi(e));
    if (const char *e = getenv("PAIS_STREAM_SHARDED")) m->streamSharded = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("PAIS_STREAM_CANARY")) m->streamCanaryMode = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("PAIS_SHARD_STRIDED")) m->shardStrided = atoi(e) != 0;
    if (const char *e = getenv("PAIS_STREAM_FIRST")) m->streamFirst = std::max(0.0, std::min(0.9, atof(e)));
    if (const char *e = getenv("PAIS_STREAM_PARTS")) m->streamParts = std::max(2, std::min(atoi(e), PAIS_MAX_STREAM_PARTS));
    if (const char *e = getenv("PAIS_STREAM_STEP")) m->streamStep = std::max(1, atoi(e));
    if (const char *e = getenv("PAIS_STREAM_SPLIT")) { const double v = atof(e); if (v > 0 && v < 1) m->streamSplit = v; }
    memset(&m->st, 0, sizeof(m->st));
    m->cfg = *cfg;
    m->cfg.patchSize = (cfg->patchRadius << 1) + 1;
    // device < 0: scheduler only (stepwise API; e.g. a rank that replays rounds but owns no GPU,
    // and the CPU tests that feed it externally computed records).  Entry points that need the
    // GPU then fail; nothing is ever computed on the host instead.
    if (device >= 0) {
        int rc = pais_ctx_create(cfg, num_cams, cams, device, pso_seed, &m->ctx);
        if (rc) { g_mvs_err = pais_last_error(); delete m; return rc; }
        m->device = device;
    }
    m->cams.resize((size_t)num_cams);
    for (int c = 0; c < num_cams; ++c) {
        const pais_camera_desc &d = cams[c];
        HostCamera &h = m->cams[c];
        h.focal[0] = d.focal[0]; h.focal[1] = d.focal[1];
        h.pp[0] = d.principle_point[0]; h.pp[1] = d.principle_point[1];
        memcpy(h.R, d.rotation, sizeof(h.R));
        memcpy(h.T, d.translation, sizeof(h.T));
        memcpy(h.C, d.center, sizeof(h.C));
        memcpy(h.optN, d.optical_normal, sizeof(h.optN));
        memcpy(h.KR, d.KR, sizeof(h.KR));
        memcpy(h.KT, d.KT, sizeof(h.KT));
        h.w0 = d.level_width[0];
        h.h0 = d.level_height[0];
        const size_t stride = d.level_stride[0] > 0 ? (size_t)d.level_stride[0] : (size_t)h.w0;
        h.img0.resize((size_t)h.w0 * h.h0);
        for (int y = 0; y < h.h0; ++y) memcpy(&h.img0[(size_t)y * h.w0], d.level_image[0] + (size_t)y * stride, (size_t)h.w0);
    }
    m->neighborRadius = cfg->neighborRadius;
    *out = m;
    return 0;
}

extern "C" void pais_mvs_destroy(pais_mvs *m) { delete m; }

// forget all patches / cell maps / queue (keeps the GPU context and the cameras): a fresh reconstruction
extern "C" int pais_mvs_reset(pais_mvs *m)
{
    if (!m) return mfail("bad argument");
    m->patchesUsed = 0; // the arena's chunks are kept for the next reconstruction
    m->patches.clear();
    m->hot.clear();
    m->alive = 0;
    m->cellMaps.clear();
    m->pool.clear();
    m->freeEntry = -1;
    m->qBest = decltype(m->qBest)();
    m->qWorst = decltype(m->qWorst)();
    m->qList.clear();
    m->qSeq = 0;
    m->liveQueued = 0;
    m->active.clear();
    m->deferred.clear();
    m->nextDeferred.clear();
    m->curRound = -1;
    m->queueExhausted = false;
    m->cands.clear();
    m->candRecs.clear();
    m->seedIds.clear();
    memset(&m->st, 0, sizeof(m->st));
    m->roundLog.clear();
    m->prevHostMs = m->prevGpuMs = 0;
    m->streamWish = m->streamAgreed = false;
    return 0;
}
extern "C" pais_ctx *pais_mvs_ctx(pais_mvs *m) { return m ? m->ctx : nullptr; }

// seed constructor: patch.cpp:26-34 + setEstimatedNormal :390-413
extern "C" int pais_mvs_add_seed(pais_mvs *m, const double center[3], int num_cam, const int32_t *cam_idx)
{
    if (!m || !center || num_cam < 0 || num_cam > PAIS_MAX_VIS || (num_cam && !cam_idx)) return mfail("pais_mvs_add_seed: bad argument");
    pais_patch_result r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < 3; ++i) r.center[i] = center[i];
    r.num_cam = num_cam;
    for (int i = 0; i < num_cam; ++i) {
        if (cam_idx[i] < 0 || cam_idx[i] >= (int)m->cams.size()) return mfail("pais_mvs_add_seed: bad camera index");
        r.cam_idx[i] = cam_idx[i];
    }
    r.type = PAIS_TYPE_SEED;
    r.fitness = DBL_MAX;
    r.priority = DBL_MAX;
    r.ref_cam = -1;
    r.lod = -1;
    r.key = (uint64_t)m->patches.size();
    if (num_cam >= m->cfg.minCamNum) {
        double normal[3] = {0, 0, 0}, dir[3];
        for (int i = 0; i < num_cam; i++) {
            const HostCamera &cam = m->cams[cam_idx[i]];
            for (int k = 0; k < 3; ++k) dir[k] = cam.C[k] - center[k];
            double sc = (1.0 / pais::norm3(dir));
            for (int k = 0; k < 3; ++k) dir[k] *= sc;
            for (int k = 0; k < 3; ++k) normal[k] += dir[k];
        }
        double sc = (1.0 / pais::norm3(normal));
        for (int k = 0; k < 3; ++k) r.normal[k] = normal[k] * sc;
        pais::normal2spherical(r.normal, r.normalS);
    } else {
        r.dropped = 1;
    }
    return m->storePatch(r);
}

// NVM seed with its image measurements (FileLoader::loadNvmPatch, fileloader.cpp:112-165: img_points are the file's
// offsets plus cols/2, rows/2) and, if `recenter`, MVS::reCentering / Patch::reCentering (mvs.cpp:134-145,
// patch.cpp:67-112): the point closest to the viewing rays through the measurements (normal equations summed over
// the cameras, solved with A.inv(DECOMP_SVD) * b), then setEstimatedNormal.  SURVEY 8(f) N4.
extern "C" int pais_mvs_add_seed_measured(pais_mvs *m, const double center[3], int num_cam, const int32_t *cam_idx,
                                          const double *img_points, int recenter)
{
    if (!m || !center || num_cam < 0 || num_cam > PAIS_MAX_VIS || (num_cam && (!cam_idx || !img_points)))
        return mfail("pais_mvs_add_seed_measured: bad argument");
    for (int i = 0; i < num_cam; ++i)
        if (cam_idx[i] < 0 || cam_idx[i] >= (int)m->cams.size()) return mfail("pais_mvs_add_seed_measured: bad camera index");
    double c[3] = {center[0], center[1], center[2]};
    if (recenter) {
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
        for (int i = 0; i < num_cam; ++i) {
            const HostCamera &cam = m->cams[cam_idx[i]];
            // pixel on the z = 1 plane of the camera, to world: R^T (p - T)
            const double p3[3] = {(img_points[2 * i] - cam.pp[0]) / cam.focal[0] - cam.T[0],
                                  (img_points[2 * i + 1] - cam.pp[1]) / cam.focal[1] - cam.T[1], 1.0 - cam.T[2]};
            double w3[3];
            for (int r = 0; r < 3; ++r) { // gemm order: k = 0..2
                double sacc = 0;
                for (int k = 0; k < 3; ++k) sacc += cam.R[k * 3 + r] * p3[k];
                w3[r] = sacc;
            }
            double n[3] = {w3[0] - cam.C[0], w3[1] - cam.C[1], w3[2] - cam.C[2]};
            const double sc = (1.0 / pais::norm3(n));
            for (int k = 0; k < 3; ++k) n[k] = n[k] * sc;
            const double *cc = cam.C;
            A[0][0] += 1 - n[0] * n[0];
            A[0][1] += -n[0] * n[1];
            A[0][2] += -n[0] * n[2];
            A[1][0] += -n[0] * n[1];
            A[1][1] += 1 - n[1] * n[1];
            A[1][2] += -n[1] * n[2];
            A[2][0] += -n[0] * n[2];
            A[2][1] += -n[1] * n[2];
            A[2][2] += 1 - n[2] * n[2];
            b[0] += (1 - n[0] * n[0]) * cc[0] - n[0] * n[1] * cc[1] - n[0] * n[2] * cc[2];
            b[1] += -n[0] * n[1] * cc[0] + (1 - n[1] * n[1]) * cc[1] - n[1] * n[2] * cc[2];
            b[2] += -n[0] * n[2] * cc[0] - n[1] * n[2] * cc[1] + (1 - n[2] * n[2]) * cc[2];
        }
        // A.inv(DECOMP_SVD): SVD back-substitution of the identity, column by column; then inv * b
        double inv[3][3];
        for (int j = 0; j < 3; ++j) {
            double Ac[3][3], e[3] = {0, 0, 0}, col[3];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) Ac[r][k] = A[r][k];
            e[j] = 1.0;
            pais::jacobi_lstsq<3, 3>(Ac, e, col);
            for (int r = 0; r < 3; ++r) inv[r][j] = col[r];
        }
        for (int r = 0; r < 3; ++r) {
            double sacc = 0;
            for (int k = 0; k < 3; ++k) sacc += inv[r][k] * b[k];
            c[r] = sacc;
        }
    }
    const int id = pais_mvs_add_seed(m, c, num_cam, cam_idx);
    if (id >= 0)
        for (int i = 0; i < num_cam; ++i) {
            m->patches[id]->r.imgPoint[i][0] = img_points[2 * i];
            m->patches[id]->r.imgPoint[i][1] = img_points[2 * i + 1];
        }
    return id;
}

// ------------------------------------------------------- sharded refinement ---
#define MHIP(call)                                                                  \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) { g_mvs_err = std::string(#call ": ") + hipGetErrorString(e__); return -2; } \
    } while (0)

// n candidates -> n records on one rank: the GPU context, or the record source of a GPU-less driver
// view != nullptr: the records may stay where they were produced (*view: valid until the next batch); else -> out
static int refine_local(pais_mvs *m, int n, const pais_candidate *c, pais_patch_result *out, int has_seeds,
                        const pais_patch_result **view = nullptr)
{
    if (n <= 0) return 0;
    if (view) *view = out;
    if (m->ctx) {
        int rc = view ? pais_refine_batch_view(m->ctx, n, c, view) : pais_refine_batch(m->ctx, n, c, out);
        if (rc) g_mvs_err = pais_last_error();
        return rc;
    }
    if (m->recordSource) {
        if (m->recordSource(m->recordUser, n, c, out, has_seeds)) return mfail("record source failed");
        return 0;
    }
    return mfail("this driver has neither a GPU context nor a record source");
}

static constexpr size_t kWireHeader = 64; // per-rank header of a sharded batch's exchange (status of the rank's refinement)
static int all_gather_host(pais_mvs *m, const void *send, void *recv, size_t bytes)
{
    if (!m->gatherCb) return mfail("no host all-gather installed");
    if (m->gatherCb(m->gatherUser, send, recv, bytes)) return mfail("all-gather callback failed");
    return 0;
}

// ------------------------------------------------------------ sharded batches ---
// A batch of n candidates over `world` ranks: contiguous count-balanced shards, every rank refines its own on its GPU
// (records stay in HBM), packs them into wire slots behind a 64-byte status header WRITTEN ON THE DEVICE
// (pais_wire_header_device: the header of a rank whose refinement failed -- or whose k_pso_ring pass did not complete -- says
// so), one ncclAllGather, one copy down, ONE host synchronisation per batch.  Every rank reads every header: all return an
// error together, or all agree on a second exchange after the ranks whose ring pass failed have refined their shard again.
// Split in submit / finish so that two batches -- the parts of a streamed round -- can be in flight: both exchanges are
// enqueued on the driver's own stream in submission order (one communicator, one stream, one order on every rank); the
// second part is refined on a lane and the exchange waits for its pack through an event.
struct WireHeader { uint32_t magic; int32_t rc; int32_t count; int32_t rank; uint32_t user; };
static bool sharded_transport(const pais_mvs *m) { return m->ctx && (m->nccl || m->emuMode == 2); }
// the HOST instance of the same protocol: blocks through the caller's all-gather (pais_mvs_comm_init_callback), this rank's shard
// refined by refine_local (its GPU context through the host batch call, or the record source of a GPU-less driver)
static bool host_transport(const pais_mvs *m) { return !sharded_transport(m) && m->gatherCb != nullptr; }

// every rank grows its buffers for the same batches (sizes follow from the replicated candidate list); a rank that cannot
// must not leave the others in the collective that follows: the outcome of a growth is agreed on first, 4 bytes per rank
static int shard_growth_handshake(pais_mvs *m, int localRc)
{
    if (host_transport(m)) { // 4 bytes per rank through the caller's all-gather
        std::vector<int> all((size_t)m->world, 0);
        int mine = localRc;
        if (all_gather_host(m, &mine, all.data(), sizeof(int))) return -3;
        for (int r = 0; r < m->world; ++r)
            if (all[(size_t)r] != 0) {
                if (!localRc) g_mvs_err = "sharded batch: rank " + std::to_string(r) + " could not grow its exchange buffers";
                return all[(size_t)r] < 0 ? all[(size_t)r] : -2;
            }
        return 0;
    }
    if (m->emuMode == 2 || !m->nccl) return localRc;
    hipStream_t xs = (hipStream_t)pais_ctx_stream(m->ctx);
    std::string err;
    rccl::Api *a = rccl::api(err);
    if (!a || !m->d_hs) return localRc ? localRc : -2;
    m->h_hs[0] = localRc;
    if (hipMemcpyAsync(m->d_hs, m->h_hs, sizeof(int), hipMemcpyHostToDevice, xs) != hipSuccess) return -2;
    if (a->allGather(m->d_hs, m->d_hsAll, sizeof(int), rccl::kInt8, m->nccl, xs) != 0) return -3;
    if (hipMemcpyAsync(m->h_hs, m->d_hsAll, sizeof(int) * (size_t)m->world, hipMemcpyDeviceToHost, xs) != hipSuccess) return -2;
    if (hipStreamSynchronize(xs) != hipSuccess) return -2;
    for (int r = 0; r < m->world; ++r)
        if (m->h_hs[r] != 0) {
            if (!localRc) g_mvs_err = "sharded batch: rank " + std::to_string(r) + " could not grow its exchange buffers";
            return m->h_hs[r] < 0 ? m->h_hs[r] : -2;
        }
    return 0;
}

static int shard_ensure(pais_mvs *m, pais_mvs::ShardBufs &B, size_t per, size_t slot)
{
    const int world = m->world;
    if (host_transport(m)) {
        const bool grow = per > B.hostC.size() || slot > B.hostS.size() || slot * (size_t)world > B.hostAll.size();
        if (!grow) return 0;
        int rc = 0;
        if (m->injectGrowthFail > 0 && --m->injectGrowthFail == 0) { rc = -2; g_mvs_err = "injected: exchange buffers could not grow"; }
        if (!rc) {
            const size_t cap = per + per / 2 + 64, capS = slot * 3 / 2 + 4096;
            B.hostC.resize(std::max(cap, B.hostC.size()));
            B.hostR.resize(B.hostC.size());
            B.hostS.resize(std::max(capS, B.hostS.size()));
            B.hostAll.resize(std::max(B.hostS.size() * (size_t)world, B.hostAll.size()));
        }
        return shard_growth_handshake(m, rc);
    }
    const bool growShard = per > B.shardCap, growWire = slot * (size_t)world > B.wireCap || slot > B.wireSCap;
    if (!growShard && !growWire && B.packed) return 0;
    hipStream_t xs = (hipStream_t)pais_ctx_stream(m->ctx);
    int rc = 0;
    auto ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) { rc = -2; g_mvs_err = std::string(what) + ": " + hipGetErrorString(e); }
        return e == hipSuccess;
    };
    ok(hipSetDevice(m->device), "hipSetDevice");
    ok(hipDeviceSynchronize(), "hipDeviceSynchronize"); // (rare: buffers of batches in flight on either lane are about to be replaced)
    if (!B.packed) {
        ok(hipEventCreateWithFlags(&B.packed, hipEventDisableTiming), "hipEventCreate");
        ok(hipEventCreateWithFlags(&B.done, hipEventDisableTiming), "hipEventCreate");
    }
    if (growShard && !rc) {
        (void)hipFree(B.d_shardC); (void)hipFree(B.d_shardR); (void)hipHostFree(B.h_shardC);
        B.d_shardC = nullptr; B.d_shardR = nullptr; B.h_shardC = nullptr; B.shardCap = 0;
        const size_t cap = per + per / 2 + 64;
        if (ok(hipMalloc((void **)&B.d_shardC, sizeof(pais_candidate) * cap), "hipMalloc(shard candidates)") &&
            ok(hipMalloc((void **)&B.d_shardR, sizeof(pais_patch_result) * cap), "hipMalloc(shard records)") &&
            ok(hipHostMalloc((void **)&B.h_shardC, sizeof(pais_candidate) * cap, hipHostMallocDefault), "hipHostMalloc(shard candidates)"))
            B.shardCap = cap;
    }
    if (growWire && !rc) {
        (void)hipFree(B.d_wireS); (void)hipFree(B.d_wireAll); (void)hipHostFree(B.h_wireAll);
        B.d_wireS = nullptr; B.d_wireAll = nullptr; B.h_wireAll = nullptr; B.wireCap = 0; B.wireSCap = 0;
        const size_t capS = slot * 3 / 2 + 4096, cap = std::max(capS * (size_t)world, B.wireCap);
        if (ok(hipMalloc(&B.d_wireS, capS), "hipMalloc(wire block)") && ok(hipMalloc(&B.d_wireAll, cap), "hipMalloc(wire blocks)") &&
            ok(hipHostMalloc((void **)&B.h_wireAll, cap, hipHostMallocDefault), "hipHostMalloc(wire blocks)")) {
            B.wireCap = cap;
            B.wireSCap = capS;
        }
    }
    (void)xs;
    return shard_growth_handshake(m, rc);
}

// which candidates of a batch of n are rank r's, and where its j-th one sits in the batch (see pais_mvs::shardStrided)
static inline int shard_count(const pais_mvs *m, int r, int n, int per)
{
    if (m->shardStrided) return r < n ? (n - r + m->world - 1) / m->world : 0;
    const int lo = std::min(r * per, n);
    return std::min(lo + per, n) - lo;
}
static inline int shard_index(const pais_mvs *m, int r, int j, int n, int per)
{
    (void)n;
    return m->shardStrided ? j * m->world + r : r * per + j;
}

// other ranks' blocks of an emulated exchange, from the records of the single-rank run (host, into the pinned staging)
static int emu_fill_blocks(pais_mvs *m, pais_mvs::ShardXfer &X)
{
    const double t0 = now_ms();
    for (int r = 0; r < m->world; ++r) {
        if (r == m->rank) continue;
        unsigned char *blk = X.B->h_wireAll + X.slot * (size_t)r;
        const int rcnt = shard_count(m, r, X.n, X.per);
        WireHeader hd = {PAIS_WIRE_MAGIC, 0, rcnt, r, m->streamWish ? 1u : 0u}; // (every emulated rank decides as this one does)
        memset(blk, 0, kWireHeader);
        memcpy(blk, &hd, sizeof(hd));
        for (int i = 0; i < rcnt; ++i) {
            auto it = m->emuIndex.find(X.c[shard_index(m, r, i, X.n, X.per)].key);
            if (it == m->emuIndex.end()) return mfail("emulated world: a candidate of this run is not among the recorded run's (another workload?)");
            if (pais_pack_records(1, &m->emuRecs[it->second], X.Kb, blk + kWireHeader + X.WB * (size_t)i)) return mfail(pais_last_error());
        }
    }
    m->st.emu_replay_ms += now_ms() - t0;
    return 0;
}

// the exchange of X's blocks, enqueued on the driver's stream behind X's pack (event), and the copy down
static int shard_exchange(pais_mvs *m, pais_mvs::ShardXfer &X)
{
    if (host_transport(m)) {
        m->st.exchange_bytes += (int64_t)(X.slot * (size_t)m->world);
        return all_gather_host(m, X.B->hostS.data(), X.B->hostAll.data(), X.slot);
    }
    hipStream_t xs = (hipStream_t)pais_ctx_stream(m->ctx);
    pais_mvs::ShardBufs &B = *X.B;
    // (a HIP error on this rank before the collective must not keep it out of the collective -- the others would wait in it
    // for ever: it is noted, the all-gather is entered regardless, and the error is returned afterwards; the garbage this rank
    // may have sent fails the header check of the others)
    hipError_t pre = hipSuccess;
    if (X.lane != m->ctx) pre = hipStreamWaitEvent(xs, B.packed, 0);
    if (m->emuMode == 2) {
        // the other ranks' blocks arrive from the host (about what the links would deliver), this rank's own from its buffer
        for (int r = 0; r < m->world; ++r) {
            unsigned char *dst = (unsigned char *)B.d_wireAll + X.slot * (size_t)r;
            if (r == m->rank) MHIP(hipMemcpyAsync(dst, B.d_wireS, X.slot, hipMemcpyDeviceToDevice, xs));
            else MHIP(hipMemcpyAsync(dst, B.h_wireAll + X.slot * (size_t)r, X.slot, hipMemcpyHostToDevice, xs));
        }
    } else {
        std::string err;
        rccl::Api *a = rccl::api(err);
        if (!a) return mfail(err.c_str());
        int nr = a->allGather(B.d_wireS, B.d_wireAll, X.slot, rccl::kInt8, m->nccl, xs);
        if (nr != 0) { g_mvs_err = std::string("ncclAllGather: ") + (a->errorString ? a->errorString(nr) : "error"); return -3; }
    }
    if (pre != hipSuccess) { g_mvs_err = std::string("hipStreamWaitEvent: ") + hipGetErrorString(pre); return -2; }
    MHIP(hipMemcpyAsync(B.h_wireAll, B.d_wireAll, X.slot * (size_t)m->world, hipMemcpyDeviceToHost, xs));
    MHIP(hipEventRecord(B.done, xs));
    m->st.exchange_bytes += (int64_t)(X.slot * (size_t)m->world);
    return 0;
}

// this rank's shard: candidates up, refined, packed, header (all on X.lane's stream, nothing waited for)
// preRc: a failure of this rank before the enqueue (hipSetDevice): nothing is refined, the header carries it to every rank
static void shard_refine_enqueue(pais_mvs *m, pais_mvs::ShardXfer &X, int preRc = 0)
{
    pais_mvs::ShardBufs &B = *X.B;
    if (host_transport(m)) {
        // host instance: the shard refined at once (refine_local), packed and headed on the host.  The injected statuses are the
        // ones the device path produces by itself: a k_pso_ring pass that did not complete (-> every rank takes a second
        // exchange), a refinement that failed (-> every rank fails the batch)
        int rc = preRc;
        if (X.cnt > 0 && !rc) {
            for (int j = 0; j < X.cnt; ++j) B.hostC[(size_t)j] = X.c[shard_index(m, m->rank, j, X.n, X.per)];
            if (m->injectRefineFail > 0 && --m->injectRefineFail == 0) rc = mfail("injected: this rank's refinement failed");
            else rc = refine_local(m, X.cnt, B.hostC.data(), B.hostR.data(), X.hasSeeds);
        }
        X.hostRetry = false;
        if (!rc && X.cnt > 0 && m->injectRingRetry > 0 && !X.secondAttempt) { --m->injectRingRetry; X.hostRetry = true; rc = PAIS_WIRE_RC_RING_RETRY; }
        memset(B.hostS.data(), 0, X.slot);
        WireHeader hd0 = {PAIS_WIRE_MAGIC, rc, X.cnt, m->rank, m->streamWish ? 1u : 0u};
        memcpy(B.hostS.data(), &hd0, sizeof(hd0));
        if ((!rc || X.hostRetry) && X.cnt > 0 && pais_pack_records(X.cnt, B.hostR.data(), X.Kb, B.hostS.data() + kWireHeader)) {
            hd0.rc = rc = -1;
            memcpy(B.hostS.data(), &hd0, sizeof(hd0));
        }
        X.localRc = X.hostRetry ? 0 : rc;
        return;
    }
    hipStream_t ls = (hipStream_t)pais_ctx_stream(X.lane);
    int rc = preRc;
    if (X.cnt > 0 && !rc) {
        for (int j = 0; j < X.cnt; ++j) B.h_shardC[j] = X.c[shard_index(m, m->rank, j, X.n, X.per)];
        if (hipMemcpyAsync(B.d_shardC, B.h_shardC, sizeof(pais_candidate) * (size_t)X.cnt, hipMemcpyHostToDevice, ls) != hipSuccess) rc = -2;
        if (!rc) rc = pais_refine_batch_device_async(X.lane, X.cnt, B.d_shardC, B.d_shardR, X.Kmax, X.hasSeeds);
        if (rc) g_mvs_err = pais_last_error();
        else if (pais_pack_records_device(X.lane, X.cnt, B.d_shardR, X.Kb, (unsigned char *)B.d_wireS + kWireHeader)) { rc = -2; g_mvs_err = pais_last_error(); }
    }
    X.localRc = rc;
    // (a rank whose refinement failed still takes part in the collective: its header carries the status)
    if (pais_wire_header_device(X.lane, m->rank, X.cnt, rc, m->streamWish ? 1u : 0u, B.d_wireS)) { if (!X.localRc) X.localRc = -2; g_mvs_err = pais_last_error(); }
    if (X.lane != m->ctx) (void)hipEventRecord(B.packed, ls);
}

// preFail != 0: this rank cannot refine its shard (e.g. no lane of its own for a part of a streamed round): it still enters the
// collective, with that status in its header, so that every rank fails the batch together
static int shard_submit(pais_mvs *m, pais_mvs::ShardXfer &X, pais_mvs::ShardBufs &B, pais_ctx *lane, int n, const pais_candidate *c, int has_seeds,
                        int preFail = 0)
{
    const int world = m->world;
    X.B = &B; X.lane = lane; X.c = c; X.n = n; X.hasSeeds = has_seeds; X.sharded = true; X.open = true; X.secondAttempt = false;
    X.per = (n + world - 1) / world;
    X.lo = std::min(m->rank * X.per, n); // (contiguous shards only)
    X.cnt = shard_count(m, m->rank, n, X.per);
    // What travels: wire slots (include/pais_hip.h "wire format of a record") sized for the batch's largest camera count --
    // the same on every rank, the candidate list is replicated
    X.Kb = 1;
    for (int i = 0; i < n; ++i) X.Kb = std::max(X.Kb, c[i].num_cam);
    X.Kmax = 1;
    for (int j = 0; j < X.cnt; ++j) X.Kmax = std::max(X.Kmax, c[shard_index(m, m->rank, j, n, X.per)].num_cam);
    X.WB = pais_record_wire_bytes(X.Kb);
    X.slot = kWireHeader + X.WB * (size_t)X.per;
    X.t0 = now_ms();
    int rc = shard_ensure(m, B, (size_t)X.per, X.slot);
    if (rc) { X.open = false; return rc; } // (agreed on by every rank: shard_growth_handshake)
    // (from here on a local failure must not keep this rank out of the collective -- the others would wait in it for ever: it
    //  travels in the status header of the rank's block and fails every rank together)
    const hipError_t sd = host_transport(m) ? hipSuccess : hipSetDevice(m->device);
    if (m->emuMode == 2 && (rc = emu_fill_blocks(m, X)) != 0) { X.open = false; return rc; }
    if (sd != hipSuccess) g_mvs_err = std::string("hipSetDevice: ") + hipGetErrorString(sd);
    shard_refine_enqueue(m, X, (sd != hipSuccess || preFail) ? -2 : 0); // (the status header written on the device says so to every rank)
    rc = shard_exchange(m, X);
    if (rc) X.open = false;
    return rc;
}

// waits for X's exchange, reads every rank's status, unpacks the n records into out
static int shard_finish(pais_mvs *m, pais_mvs::ShardXfer &X, pais_patch_result *out)
{
    if (!X.open) return mfail("sharded batch: nothing in flight");
    X.open = false;
    const int world = m->world;
    pais_mvs::ShardBufs &B = *X.B;
    const bool host = host_transport(m);
    const unsigned char *wireAll = host ? B.hostAll.data() : B.h_wireAll;
    for (int attempt = 0;; ++attempt) {
        if (!host) MHIP(hipEventSynchronize(B.done));
        if (m->emuMode == 2 && m->emuLatencyUs > 0) { // modelled launch latency of the collective (the bytes moved for real, over PCIe)
            const double t1 = now_ms() + m->emuLatencyUs * 1e-3;
            while (now_ms() < t1) {}
        }
        const int mine = host ? (X.hostRetry ? 1 : 0)
                              : (X.cnt > 0 ? pais_ctx_batch_status(X.lane) : 0); // (consumes this rank's ring status; the header says the same)
        bool retry = false;
        for (int r = 0; r < world; ++r) {
            WireHeader hd;
            memcpy(&hd, wireAll + X.slot * (size_t)r, sizeof(hd));
            if (r == 0 && hd.magic == PAIS_WIRE_MAGIC) m->streamAgreed = (hd.user & 1u) != 0; // rank 0's choice binds every rank
            const int rcnt = shard_count(m, r, X.n, X.per);
            if (hd.magic != PAIS_WIRE_MAGIC || hd.rank != r || hd.count != rcnt) return mfail("sharded batch: malformed exchange header (ranks disagree on the batch)");
            if (hd.rc == PAIS_WIRE_RC_RING_RETRY) { retry = true; continue; }
            if (hd.rc != 0) {
                if (r != m->rank || g_mvs_err.empty()) g_mvs_err = "sharded batch: rank " + std::to_string(r) + " failed to refine its shard (rc " + std::to_string(hd.rc) + ")";
                return hd.rc < 0 ? hd.rc : -1;
            }
        }
        if (!retry) break;
        if (attempt >= 1) return mfail("sharded batch: a shard did not complete twice");
        // some rank's k_pso_ring pass did not complete: those ranks refine their shard again (one launch per iteration), every
        // rank sends its block again -- all ranks have read the same headers, so all take this second exchange
        X.secondAttempt = true;
        m->st.exchange_retries++;
        if (mine == 1) shard_refine_enqueue(m, X);
        else if (!host && X.lane != m->ctx) MHIP(hipEventRecord(B.packed, (hipStream_t)pais_ctx_stream(X.lane)));
        int rc = shard_exchange(m, X);
        if (rc) return rc;
    }
    m->st.exchange_ms += now_ms() - X.t0; // (from the submission: refinement of the shard included)
    for (int r = 0; r < world; ++r) {
        const int rcnt = shard_count(m, r, X.n, X.per);
        const unsigned char *blk = wireAll + X.slot * (size_t)r + kWireHeader;
        if (!m->shardStrided) {
            if (rcnt > 0 && pais_unpack_records(rcnt, blk, X.Kb, out + shard_index(m, r, 0, X.n, X.per))) return mfail(pais_last_error());
        } else {
            for (int j = 0; j < rcnt; ++j)
                if (pais_unpack_records(1, blk + X.WB * (size_t)j, X.Kb, out + shard_index(m, r, j, X.n, X.per))) return mfail(pais_last_error());
        }
    }
    return 0;
}

static void emu_record(pais_mvs *m, int n, const pais_candidate *c, const pais_patch_result *recs)
{
    for (int i = 0; i < n; ++i) {
        auto it = m->emuIndex.find(c[i].key);
        if (it != m->emuIndex.end()) { m->emuRecs[it->second] = recs[i]; continue; }
        m->emuIndex.emplace(c[i].key, (uint32_t)m->emuRecs.size());
        m->emuRecs.push_back(recs[i]);
    }
}

// The batch entry of the drivers: one rank -> pais_refine_batch; several ranks -> shard, refine, all-gather.
// *view: where the n records are (out, or a pinned staging buffer that stays valid until the next batch)
static int refine_any(pais_mvs *m, int n, const pais_candidate *c, pais_patch_result *out, int has_seeds, const pais_patch_result **view)
{
    *view = out;
    if (n <= 0) return 0;
    const int world = m->world;
    if (world <= 1 && !m->nccl && !m->gatherCb) {
        const int rc = refine_local(m, n, c, out, has_seeds, view);
        if (!rc && m->emuMode == 1) emu_record(m, n, c, *view);
        return rc;
    }
    // a batch of fewer than replicateBelow evaluation waves per PSO iteration (candidates x particles; seeds run twice the
    // particles) is latency bound on ONE GPU: its per-iteration launches take one evaluation wave's latency whatever the
    // number of GPUs, so splitting it buys nothing and the exchange costs -- replicated, no collective
    const long wavesPerIter = (long)n * m->cfg.particleNum * (has_seeds ? 2 : 1);
    if (wavesPerIter < (long)m->replicateBelow) {
        m->st.batches_replicated++;
        return refine_local(m, n, c, out, has_seeds, view);
    }
    m->st.batches_sharded++;
    m->results.resize((size_t)n);
    // RCCL (or its one-GPU emulation): records stay in HBM, ONE ncclAllGather on the driver's stream, ONE synchronisation.
    // A caller-supplied all-gather / a GPU-less driver: the same submit / finish protocol over host memory (host_transport).
    if (!sharded_transport(m) && !host_transport(m)) return mfail("sharded batch: no transport attached");
    int rc = shard_submit(m, m->sx[0], m->sb[0], m->ctx, n, c, has_seeds);
    if (!rc) rc = shard_finish(m, m->sx[0], m->results.data());
    if (rc) return rc;
    *view = m->results.data();
    return 0;
}

extern "C" int pais_mvs_test_inject(pais_mvs *m, int what, int count)
{
    if (!m || count < 0) return mfail("pais_mvs_test_inject: bad argument");
    if (what == 1) m->injectRingRetry = count;
    else if (what == 2) m->injectGrowthFail = count;
    else if (what == 3) m->injectRefineFail = count;
    else return mfail("pais_mvs_test_inject: unknown injection");
    return 0;
}

extern "C" int pais_comm_get_unique_id(pais_unique_id *out)
{
    if (!out) return mfail("pais_comm_get_unique_id: bad argument");
    std::string err;
    rccl::Api *a = rccl::api(err);
    if (!a) return mfail(err.c_str());
    rccl::UniqueId id;
    int nr = a->getUniqueId(&id);
    if (nr != 0) { g_mvs_err = std::string("ncclGetUniqueId: ") + (a->errorString ? a->errorString(nr) : "error"); return -3; }
    memcpy(out->bytes, id.internal, PAIS_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int pais_mvs_comm_init_rccl(pais_mvs *m, int rank, int world, const pais_unique_id *id)
{
    if (!m || !id || world < 1 || rank < 0 || rank >= world) return mfail("pais_mvs_comm_init_rccl: bad argument");
    if (!m->ctx) return mfail("pais_mvs_comm_init_rccl: this driver owns no GPU");
    if (m->nccl || m->gatherCb) return mfail("pais_mvs_comm_init_rccl: a communicator is already attached");
    std::string err;
    rccl::Api *a = rccl::api(err);
    if (!a) return mfail(err.c_str());
    MHIP(hipSetDevice(m->device));
    rccl::UniqueId uid;
    memcpy(uid.internal, id->bytes, PAIS_UNIQUE_ID_BYTES);
    rccl::Comm comm = nullptr;
    int nr = a->commInitRank(&comm, world, uid, rank);
    if (nr != 0) { g_mvs_err = std::string("ncclCommInitRank: ") + (a->errorString ? a->errorString(nr) : "error"); return -3; }
    m->nccl = comm;
    m->rank = rank;
    m->world = world;
    MHIP(hipMalloc((void **)&m->d_hs, sizeof(int)));
    MHIP(hipMalloc((void **)&m->d_hsAll, sizeof(int) * (size_t)world));
    MHIP(hipHostMalloc((void **)&m->h_hs, sizeof(int) * (size_t)std::max(world, 1), hipHostMallocDefault));
    return 0;
}

// One-GPU emulation of one rank of a larger world (measurement aid: bench.py --emulate-world).
//   mode 1 (record; a single-rank driver): the records of every batch are kept, keyed by candidate (they survive pais_mvs_reset);
//   mode 2 (emulate rank `rank` of `world`): every sharded batch runs the real sharded code path -- this rank's shard refined on
//           the GPU, packed, header, copy down, unpack, replicated commit, thin batches replicated -- with the other ranks' blocks
//           packed from the recorded records and copied to the device in place of the ncclAllGather (plus PAIS_EMU_LATENCY_US
//           of modelled collective launch latency); the host time spent preparing those blocks is accounted in
//           pais_mvs_stats::emu_replay_ms, work a real rank does not have;
//   mode 0: off (the recorded records are dropped).
extern "C" int pais_mvs_emulate(pais_mvs *m, int mode, int rank, int world)
{
    if (!m || mode < 0 || mode > 2) return mfail("pais_mvs_emulate: bad argument");
    if (m->nccl || m->gatherCb) return mfail("pais_mvs_emulate: a communicator is attached");
    if (mode == 2 && (!m->ctx || world < 1 || rank < 0 || rank >= world)) return mfail("pais_mvs_emulate: bad rank / world, or no GPU");
    if (mode == 2 && m->emuRecs.empty()) return mfail("pais_mvs_emulate: nothing recorded (run the workload in mode 1 first)");
    if (const char *e = getenv("PAIS_EMU_LATENCY_US")) m->emuLatencyUs = atof(e);
    m->emuMode = mode;
    if (mode == 2) { m->rank = rank; m->world = world; }
    else { m->rank = 0; m->world = 1; }
    if (mode == 0) { m->emuIndex.clear(); std::vector<pais_patch_result>().swap(m->emuRecs); }
    return 0;
}

extern "C" int pais_mvs_create_ranked(const pais_config *cfg, int num_cams, const pais_camera_desc *cams, int device,
                                      uint64_t pso_seed, int rank, int world, const pais_unique_id *id, pais_mvs **out)
{
    if (device < 0) return mfail("pais_mvs_create_ranked: a rank needs a GPU");
    int rc = pais_mvs_create(cfg, num_cams, cams, device, pso_seed, out);
    if (rc) return rc;
    rc = pais_mvs_comm_init_rccl(*out, rank, world, id);
    if (rc) { pais_mvs_destroy(*out); *out = nullptr; }
    return rc;
}

extern "C" int pais_mvs_comm_init_callback(pais_mvs *m, int rank, int world, pais_allgather_fn fn, void *user)
{
    if (!m || !fn || world < 1 || rank < 0 || rank >= world) return mfail("pais_mvs_comm_init_callback: bad argument");
    if (m->nccl || m->gatherCb) return mfail("pais_mvs_comm_init_callback: a communicator is already attached");
    m->gatherCb = fn;
    m->gatherUser = user;
    m->rank = rank;
    m->world = world;
    return 0;
}

extern "C" int pais_mvs_set_replicate_below(pais_mvs *m, int waves)
{
    if (!m || waves < 0) return mfail("pais_mvs_set_replicate_below: bad argument");
    m->replicateBelow = waves;
    return 0;
}

extern "C" int pais_mvs_set_record_source(pais_mvs *m, pais_record_source_fn fn, void *user)
{
    if (!m) return mfail("pais_mvs_set_record_source: bad argument");
    if (m->ctx) return mfail("pais_mvs_set_record_source: this driver owns a GPU context; records come from its kernels only");
    m->recordSource = fn;
    m->recordUser = user;
    return 0;
}

extern "C" int pais_mvs_seed_begin(pais_mvs *m, const pais_candidate **cands, int *n)
{
    if (!m || !cands || !n) return mfail("pais_mvs_seed_begin: bad argument");
    if (m->alive == 0) { *cands = nullptr; *n = 0; return 0; }
    m->setNeighborRadius(); // mvs.cpp:202
    m->candRecs.clear();
    m->seedIds.clear();
    for (size_t id = 0; id < m->patches.size(); ++id) {
        HostPatch *p = m->patches[id];
        if (!p) continue;
        if (p->r.num_cam < m->cfg.minCamNum) { m->deletePatch((int)id); continue; } // :209-212
        pais_candidate c;
        memset(&c, 0, sizeof(c));
        for (int i = 0; i < 3; ++i) { c.center[i] = p->r.center[i]; c.normal[i] = p->r.normal[i]; }
        c.normalS[0] = p->r.normalS[0]; c.normalS[1] = p->r.normalS[1];
        c.key = p->r.key;
        c.type = PAIS_TYPE_SEED;
        c.num_cam = p->r.num_cam;
        for (int i = 0; i < c.num_cam; ++i) c.cam_idx[i] = p->r.cam_idx[i];
        m->candRecs.push_back(c);
        m->seedIds.push_back((int)id);
    }
    *cands = m->candRecs.data();
    *n = (int)m->candRecs.size();
    return 0;
}

extern "C" int pais_mvs_seed_commit(pais_mvs *m, const pais_patch_result *results, int n)
{
    if (!m || n != (int)m->seedIds.size() || (n && !results)) return mfail("pais_mvs_seed_commit: bad argument");
    for (int k = 0; k < n; ++k) {
        const int id = m->seedIds[k];
        HostPatch *p = m->patches[id];
        p->r = results[k];
        p->childCams = -1;
        {
            HotPatch &h = m->hot[id];
            for (int i = 0; i < 3; ++i) { h.center[i] = p->r.center[i]; h.normal[i] = p->r.normal[i]; }
            h.correlation = p->r.correlation;
            if (!m->cellMaps.empty()) m->blocksValid = false; // (maps built before this refinement: rebuilt by pais_mvs_expansion_begin)
        }
        m->st.seeds_refined++;
        m->st.pso_evals_effective += results[k].pso_evals;
        if (!m->runtimeFiltering(p->r, id, m->sceneKnowledge(p->r))) m->deletePatch(id); // :217-220
        else p->inScene = true;
    }
    m->setNeighborRadius(); // :230
    m->seedIds.clear();
    return 0;
}

extern "C" int pais_mvs_refine_seed_patches(pais_mvs *m)
{
    const pais_candidate *c;
    int n;
    int rc = pais_mvs_seed_begin(m, &c, &n);
    if (rc) return rc;
    if (n == 0) return 0;
    if (!m->ctx && !m->recordSource) return mfail("pais_mvs_refine_seed_patches: this driver was created without a GPU context");
    m->results.resize((size_t)n);
    double t0 = now_ms();
    const pais_patch_result *recs = nullptr;
    const int64_t shardedBefore = m->st.batches_sharded;
    rc = refine_any(m, n, c, m->results.data(), 1, &recs);
    const double tRef = now_ms() - t0;
    m->st.gpu_refine_ms += tRef;
    if (rc) return rc;
    int kmax = 1;
    for (int i = 0; i < n; ++i) kmax = std::max(kmax, c[i].num_cam);
    const double t1 = now_ms();
    rc = pais_mvs_seed_commit(m, recs, n);
    m->roundLog.push_back(pais_round_log{n, 1, m->st.batches_sharded > shardedBefore ? 1 : 0, kmax, tRef, 0.0, now_ms() - t1});
    return rc;
}

extern "C" int pais_mvs_expansion_begin(pais_mvs *m)
{
    if (!m) return mfail("bad argument");
    m->setCellMaps();
    m->initPriorityQueue();
    m->setNeighborRadius();
    m->active.clear();
    m->deferred.clear();
    m->nextDeferred.clear();
    m->curRound = 0;
    m->queueExhausted = false;
    return 0;
}

// One round of the schedule R(B) (the oracle's po_mvs_expansion_patches
// states the same schedule one candidate at a time):
//   1. top the ordered active set up to B parents from the queue (reference pop policy,
//      setExpanded, runtimeFiltering/delete: mvs.cpp:245-260);
//   2. work list = [units deferred by the previous round] + the <= 4 neighbour cells of the CURRENT
//      camera slot of every active parent; a unit blocked by skipNeighborCell on the pre-round state
//      is dropped; of the units that target one (camera, cell) only the first is taken ("claims" the
//      cell), the others are deferred to the next round;
//   3. (caller) the claimed units are refined in ONE GPU batch -- a superset of what the sequential
//      order evaluates, because insertions can only turn a candidate into a skip;
//   4. round_commit replays the sequential order with the skip test re-applied on the live state.
// The exchange of a sharded listing: every rank holds the states (0 dropped or not mine, 1 deferred, 2 claimed) of the units
// of ITS (camera, tile) buckets; afterwards every rank holds every unit's state (the maximum over the ranks: exactly one
// rank owns a unit).  Host transport: the caller's all-gather; RCCL: up, ncclAllGather, down, one synchronisation.
static int enum_exchange(pais_mvs *m, std::vector<unsigned char> &st)
{
    const size_t n = st.size(), W = (size_t)m->world;
    if (n == 0) return 0;
    const double t0 = now_ms();
    const unsigned char *all = nullptr;
    if (host_transport(m)) {
        m->enumAll.resize(n * W);
        if (all_gather_host(m, st.data(), m->enumAll.data(), n)) return -3;
        all = m->enumAll.data();
    } else {
        if (!m->nccl) return mfail("sharded enumeration: no communicator");
        hipStream_t xs = (hipStream_t)pais_ctx_stream(m->ctx);
        std::string err;
        rccl::Api *a = rccl::api(err);
        if (!a) return mfail(err.c_str());
        if (n > m->enCap) {
            MHIP(hipStreamSynchronize(xs));
            (void)hipFree(m->d_en); (void)hipFree(m->d_enAll); (void)hipHostFree(m->h_en);
            m->d_en = m->d_enAll = m->h_en = nullptr;
            m->enCap = n + n / 2 + 4096; // (grows for the same rounds on every rank: the unit count is replicated state)
            MHIP(hipMalloc((void **)&m->d_en, m->enCap));
            MHIP(hipMalloc((void **)&m->d_enAll, m->enCap * W));
            MHIP(hipHostMalloc((void **)&m->h_en, m->enCap * (W + 1), hipHostMallocDefault));
        }
        memcpy(m->h_en, st.data(), n);
        MHIP(hipMemcpyAsync(m->d_en, m->h_en, n, hipMemcpyHostToDevice, xs));
        const int nr = a->allGather(m->d_en, m->d_enAll, n, rccl::kInt8, m->nccl, xs);
        if (nr != 0) { g_mvs_err = std::string("ncclAllGather (unit states): ") + (a->errorString ? a->errorString(nr) : "error"); return -3; }
        MHIP(hipMemcpyAsync(m->h_en + m->enCap, m->d_enAll, n * W, hipMemcpyDeviceToHost, xs));
        MHIP(hipStreamSynchronize(xs));
        all = m->h_en + m->enCap;
    }
    for (size_t r = 0; r < W; ++r) {
        const unsigned char *blk = all + r * n;
        for (size_t k = 0; k < n; ++k) st[k] = std::max(st[k], blk[k]);
    }
    m->st.exchange_ms += now_ms() - t0;
    m->st.exchange_bytes += (int64_t)(n * W);
    return 0;
}

extern "C" int pais_mvs_round_begin(pais_mvs *m, int B, const pais_candidate **cands, int *n)
{
    if (!m || !cands || !n) return mfail("pais_mvs_round_begin: bad argument");
    double t0 = now_ms();
    if (B < 1) B = 1;
    *cands = nullptr;
    *n = 0;
    m->firstPart = -1; // (set below if this round's first part is handed out early)
    while ((int)m->active.size() < B && !m->queueExhausted) {
        int id = m->queuePop();
        if (id < 0) break;
        // reference quirk (mvs.cpp:241-243,271): with one parent at a time, the parent popped
        // last is expanded only while the queue is still non-empty
        if (B == 1 && m->strictTail && m->liveQueued == 0) { m->queueExhausted = true; break; }
        HostPatch *p = m->patches[id];
        p->expanded = true;                                   // :250
        m->st.parents_popped++;
        // (a patch kept by the runtimeFiltering of its insertion: the camera loop cannot answer otherwise now)
        if (!m->runtimeFiltering(p->r, p->id, (m->trustSceneStage && p->inScene) ? 1 : 0)) { m->deletePatch(id); continue; } // :255-260
        m->active.push_back(Active{id, 0});
    }
    if (m->active.empty() && m->deferred.empty()) return 1;

    m->cands.clear();
    m->candRecs.clear();
    m->nextDeferred.clear();
    // one neighbour cell (x, y) of camera camI of the parent of unit u
    auto considerCell = [&](const Unit &u, const pais_patch_result &pr, int camI, const CellMap &map, int x, int y) {
        if (!map.inMap(x, y)) return;
        if (m->skipNeighborCell(map, x, y, pr, m->curRound)) return; // blocked before the round (== live: nothing inserted yet)
        if (!m->cellMaps[camI].claim(x, y, m->curRound)) { m->nextDeferred.push_back(u); return; } // one attempt per cell per round
        double center[3];
        m->expansionCenter(camI, pr, x, y, center);
        pais_candidate rec;
        m->makeExpandCandidate(m->patches[u.id], center, pais_child_key(pr.key, camI, x, y), &rec);
        m->cands.push_back(Candidate{u, camI, x, y});
        m->candRecs.push_back(rec);
    };
    static const int dx[4] = {-1, 0, 1, 0}, dy[4] = {0, -1, 0, 1}; // neighbour j of a cell: left, up, right, down
    // one camera slot of every active parent; a thin front (few active parents: the long tail of the
    // expansion, where a round is pure latency) takes all remaining slots of its parents at once
    const bool thin = (int)m->active.size() <= m->thinFront;
    // Large rounds on several host threads.  What a unit does -- dropped (cell blocked before the round), deferred (cell
    // claimed by an earlier unit of the round) or claimed -- depends on the pre-round state and on the units that aim at the
    // SAME cell before it in work-list order, nothing else.  So: (1) one thread lists the units in order (the parents'
    // records are the only thing it reads) and makes sure the 32 x 32 tile of every target cell exists; (2) the units are
    // dealt to the threads by (camera, tile) -- all units of a cell to one thread, each thread in work-list order -- and every
    // thread runs the skip test and the claim for its units (the cell maps, the pool and the dense patch copies are only
    // read; a claim stamp belongs to its tile's thread); (3) one thread walks the states in order and builds the candidates.
    // Same candidates, same order, same deferred list as the single-thread walk below (tests/test_scheduler_cpu.py runs both).
    // Sharded listing (opt-in): the same three phases as the threaded walk below, with the world's ranks in the place of the
    // threads -- (1) every rank lists every unit in order, (2) rank r runs the skip test and the claim for the units of ITS
    // (camera, tile) buckets, the states are exchanged, (3) every rank builds the candidates of the claimed units in order.
    // Same candidates, same order, same deferred list as the single-rank walk (tests/test_distributed_cpu.py).
    if (m->shardEnum && m->world > 1 && !m->onFirstPart && (host_transport(m) || m->nccl || m->emuMode == 2) &&
        m->deferred.size() + 4 * m->active.size() >= m->shardEnumAbove) {
        const uint32_t W = (uint32_t)m->world;
        m->partEnd.clear();
        m->work.clear();
        m->workBucket.resize((size_t)W);
        for (auto &b : m->workBucket) b.clear();
        auto addUnit = [&](const Unit &u, int camI, int x, int y) {
            CellMap &map = m->cellMaps[camI];
            if (!map.inMap(x, y)) return;
            (void)map.slot(x, y); // (the tile exists on every rank alike, whoever owns the unit)
            const uint32_t key = (uint32_t)camI * 0x9E3779B1u + (uint32_t)(y >> CellMap::kShift) * 0x85EBCA6Bu + (uint32_t)(x >> CellMap::kShift) * 0xC2B2AE35u;
            m->workBucket[(key >> 8) % W].push_back((uint32_t)m->work.size());
            m->work.push_back(WorkItem{u, camI, x, y});
        };
        for (const Unit &u : m->deferred) {
            const pais_patch_result &pr = m->patches[u.id]->r;
            addUnit(u, pr.cam_idx[u.slot], (int)(pr.imgPoint[u.slot][0] / m->cfg.cellSize) + dx[u.j], (int)(pr.imgPoint[u.slot][1] / m->cfg.cellSize) + dy[u.j]);
        }
        const size_t nA = m->active.size();
        for (size_t k = 0; k < nA; ++k) {
            if (k + 16 < nA) {
                const pais_patch_result &q = m->patches[m->active[k + 16].id]->r;
                __builtin_prefetch(&q.imgPoint[m->active[k + 16].slot][0]);
                __builtin_prefetch(&q.cam_idx[m->active[k + 16].slot]);
            }
            Active &a = m->active[k];
            const pais_patch_result &pr = m->patches[a.id]->r;
            const int sEnd = thin ? pr.num_cam : a.slot + 1;
            for (int sl = a.slot; sl < sEnd; ++sl) {
                const int cx = (int)(pr.imgPoint[sl][0] / m->cfg.cellSize), cy = (int)(pr.imgPoint[sl][1] / m->cfg.cellSize);
                for (int j = 0; j < 4; ++j) addUnit(Unit{a.id, sl, j}, pr.cam_idx[sl], cx + dx[j], cy + dy[j]);
            }
            a.slot = sEnd - 1; // round_commit advances past it
        }
        m->workState.assign(m->work.size(), 0);
        const int round = m->curRound;
        auto runBucket = [&](uint32_t w) {
            const std::vector<uint32_t> &mine = m->workBucket[(size_t)w];
            const size_t nMine = mine.size();
            for (size_t q = 0; q < nMine; ++q) {
                if (q + 8 < nMine) { // the tile row of the unit 8 ahead, the first pool entry of the unit 4 ahead
                    const WorkItem &f = m->work[mine[q + 8]];
                    m->cellMaps[f.cam].prefetch(f.x, f.y);
                    const WorkItem &g = m->work[mine[q + 4]];
                    const int e = m->cellMaps[g.cam].first(g.x, g.y);
                    if (e >= 0) __builtin_prefetch(&m->pool[e]);
                }
                const WorkItem &it = m->work[mine[q]];
                const pais_patch_result &pr = m->patches[it.u.id]->r;
                CellMap &map = m->cellMaps[it.cam];
                if (m->skipNeighborCell(map, it.x, it.y, pr, round)) continue;                  // state 0: dropped
                m->workState[mine[q]] = map.claim(it.x, it.y, round) ? 2 : 1;                   // 2 claimed, 1 deferred
            }
        };
        runBucket((uint32_t)m->rank);
        if (m->emuMode == 2) {
            // one-GPU emulation: there are no peers -- their buckets are walked here, off the emulated rank's clock (emu_replay_ms),
            // and the exchange is charged with the modelled latency of a collective
            const double tE = now_ms();
            for (uint32_t w = 0; w < W; ++w)
                if (w != (uint32_t)m->rank) runBucket(w);
            m->st.emu_replay_ms += now_ms() - tE;
            if (m->emuLatencyUs > 0) {
                const double t1 = now_ms() + m->emuLatencyUs * 1e-3;
                while (now_ms() < t1) {}
                m->st.exchange_ms += m->emuLatencyUs * 1e-3;
            }
        } else if (enum_exchange(m, m->workState)) {
            return -2;
        }
        m->st.rounds_enum_sharded++;
        for (size_t k = 0; k < m->work.size(); ++k) {
            const WorkItem &it = m->work[k];
            if (m->workState[k] == 1) { m->nextDeferred.push_back(it.u); continue; }
            if (m->workState[k] != 2) continue;
            const pais_patch_result &pr = m->patches[it.u.id]->r;
            double center[3];
            m->expansionCenter(it.cam, pr, it.x, it.y, center);
            pais_candidate rec;
            m->makeExpandCandidate(m->patches[it.u.id], center, pais_child_key(pr.key, it.cam, it.x, it.y), &rec);
            m->cands.push_back(Candidate{it.u, it.cam, it.x, it.y});
            m->candRecs.push_back(rec);
        }
        if (m->truncatedVisible > 0) {
            m->truncatedVisible = 0;
            return mfail("a candidate's visibility cone holds more than PAIS_MAX_VIS cameras (patch.cpp:723-761 keeps them all): "
                         "this rig needs a larger PAIS_MAX_VIS or a larger visibleCorrelation");
        }
        *cands = m->candRecs.data();
        *n = (int)m->candRecs.size();
        m->lastEnumerateMs = now_ms() - t0;
        m->st.host_enumerate_ms += m->lastEnumerateMs;
        return 0;
    }
    if (m->enumThreads > 1 && m->deferred.size() + 4 * m->active.size() >= m->enumThreadsAbove) {
        if (!m->enumPool) m->enumPool = new EnumPool(m->enumThreads - 1);
        const int T = m->enumPool->workers();
        m->work.clear();
        m->workBucket.resize((size_t)T);
        for (auto &b : m->workBucket) b.clear();
        auto addUnit = [&](const Unit &u, int camI, int x, int y) {
            CellMap &map = m->cellMaps[camI];
            if (!map.inMap(x, y)) return;
            (void)map.slot(x, y); // the tile exists from here on: no thread allocates
            const uint32_t key = (uint32_t)camI * 0x9E3779B1u + (uint32_t)(y >> CellMap::kShift) * 0x85EBCA6Bu + (uint32_t)(x >> CellMap::kShift) * 0xC2B2AE35u;
            m->workBucket[(key >> 8) % (uint32_t)T].push_back((uint32_t)m->work.size());
            m->work.push_back(WorkItem{u, camI, x, y});
        };
        for (const Unit &u : m->deferred) {
            const pais_patch_result &pr = m->patches[u.id]->r;
            addUnit(u, pr.cam_idx[u.slot], (int)(pr.imgPoint[u.slot][0] / m->cfg.cellSize) + dx[u.j], (int)(pr.imgPoint[u.slot][1] / m->cfg.cellSize) + dy[u.j]);
        }
        const size_t nA = m->active.size();
        for (size_t k = 0; k < nA; ++k) {
            if (k + 16 < nA) {
                const pais_patch_result &q = m->patches[m->active[k + 16].id]->r;
                __builtin_prefetch(&q.imgPoint[m->active[k + 16].slot][0]);
                __builtin_prefetch(&q.cam_idx[m->active[k + 16].slot]);
            }
            Active &a = m->active[k];
            const pais_patch_result &pr = m->patches[a.id]->r;
            const int sEnd = thin ? pr.num_cam : a.slot + 1;
            for (int sl = a.slot; sl < sEnd; ++sl) {
                const int cx = (int)(pr.imgPoint[sl][0] / m->cfg.cellSize), cy = (int)(pr.imgPoint[sl][1] / m->cfg.cellSize);
                for (int j = 0; j < 4; ++j) addUnit(Unit{a.id, sl, j}, pr.cam_idx[sl], cx + dx[j], cy + dy[j]);
            }
            a.slot = sEnd - 1; // round_commit advances past it
        }
        m->workState.assign(m->work.size(), 0);
        const int round = m->curRound;
        std::function<void(int)> job = [&](int w) {
            const std::vector<uint32_t> &mine = m->workBucket[(size_t)w];
            const size_t nMine = mine.size();
            for (size_t q = 0; q < nMine; ++q) {
                if (q + 8 < nMine) { // the tile row of the unit 8 ahead, the first pool entry of the unit 4 ahead
                    const WorkItem &f = m->work[mine[q + 8]];
                    m->cellMaps[f.cam].prefetch(f.x, f.y);
                    const WorkItem &g = m->work[mine[q + 4]];
                    const int e = m->cellMaps[g.cam].first(g.x, g.y);
                    if (e >= 0) __builtin_prefetch(&m->pool[e]);
                }
                const WorkItem &it = m->work[mine[q]];
                const pais_patch_result &pr = m->patches[it.u.id]->r;
                CellMap &map = m->cellMaps[it.cam];
                if (m->skipNeighborCell(map, it.x, it.y, pr, round)) continue;                  // state 0: dropped
                m->workState[mine[q]] = map.claim(it.x, it.y, round) ? 2 : 1;                   // 2 claimed, 1 deferred
            }
        };
        m->enumPool->run(job);
        for (size_t k = 0; k < m->work.size(); ++k) {
            const WorkItem &it = m->work[k];
            if (m->workState[k] == 1) { m->nextDeferred.push_back(it.u); continue; }
            if (m->workState[k] != 2) continue;
            const pais_patch_result &pr = m->patches[it.u.id]->r;
            double center[3];
            m->expansionCenter(it.cam, pr, it.x, it.y, center);
            pais_candidate rec;
            m->makeExpandCandidate(m->patches[it.u.id], center, pais_child_key(pr.key, it.cam, it.x, it.y), &rec);
            m->cands.push_back(Candidate{it.u, it.cam, it.x, it.y});
            m->candRecs.push_back(rec);
        }
        if (m->truncatedVisible > 0) {
            m->truncatedVisible = 0;
            return mfail("a candidate's visibility cone holds more than PAIS_MAX_VIS cameras (patch.cpp:723-761 keeps them all): "
                         "this rig needs a larger PAIS_MAX_VIS or a larger visibleCorrelation");
        }
        *cands = m->candRecs.data();
        *n = (int)m->candRecs.size();
        m->lastEnumerateMs = now_ms() - t0;
        m->st.host_enumerate_ms += m->lastEnumerateMs;
        return 0;
    }
    for (const Unit &u : m->deferred) {
        const pais_patch_result &pr = m->patches[u.id]->r;
        const int camI = pr.cam_idx[u.slot];
        const int cx = (int)(pr.imgPoint[u.slot][0] / m->cfg.cellSize);
        const int cy = (int)(pr.imgPoint[u.slot][1] / m->cfg.cellSize);
        considerCell(u, pr, camI, m->cellMaps[camI], cx + dx[u.j], cy + dy[u.j]);
    }
    // The walk is a chain of dependent cache misses (parent record -> tile table -> tile row -> pool entry -> hot patch) on
    // maps far larger than the caches (ring: 66 MB of cell heads): the records of the parents 16 ahead and the three tile
    // rows that the parent 8 ahead will look at are requested while the current one is processed.
    const size_t nAct = m->active.size();
    auto prefetchRecord = [&](size_t k) {
        if (k >= nAct) return;
        const pais_patch_result &q = m->patches[m->active[k].id]->r;
        __builtin_prefetch(&q.imgPoint[m->active[k].slot][0]);
        __builtin_prefetch(&q.cam_idx[m->active[k].slot]);
        __builtin_prefetch(&q.center[0]);
    };
    auto prefetchCells = [&](size_t k) {
        if (k >= nAct) return;
        const Active &b = m->active[k];
        const pais_patch_result &q = m->patches[b.id]->r;
        const CellMap &map = m->cellMaps[q.cam_idx[b.slot]];
        const int cx = (int)(q.imgPoint[b.slot][0] / m->cfg.cellSize), cy = (int)(q.imgPoint[b.slot][1] / m->cfg.cellSize);
        for (int dyy = -1; dyy <= 1; ++dyy)
            if (map.inMap(cx, cy + dyy)) map.prefetch(cx, cy + dyy);
    };
    // ... and, where the maps exceed the last-level cache by far (deepPrefetch: > 16 MB of cell heads; measured ring -9 %
    // enumerate time, pawn +0.9 ms if done there too), once those rows have arrived, the first pool entry of the four cells
    // (4 ahead) and its patch (2 ahead)
    auto prefetchEntries = [&](size_t k, bool patch) {
        if (k >= nAct) return;
        const Active &b = m->active[k];
        const pais_patch_result &q = m->patches[b.id]->r;
        const CellMap &map = m->cellMaps[q.cam_idx[b.slot]];
        const int cx = (int)(q.imgPoint[b.slot][0] / m->cfg.cellSize), cy = (int)(q.imgPoint[b.slot][1] / m->cfg.cellSize);
        for (int j = 0; j < 4; ++j) {
            const int x = cx + dx[j], y = cy + dy[j];
            if (!map.inMap(x, y)) continue;
            const CellMap::Cell *cl = map.cell(x, y);
            if (!cl || cl->head < 0) continue;
            if (m->blocksValid && cl->block < m->curRound) continue; // (answered from the tile row: the list is not walked)
            const int e = cl->head;
            if (patch) __builtin_prefetch(&m->hot[m->pool[e].id]);
            else __builtin_prefetch(&m->pool[e]);
        }
    };
    for (size_t k = 0; k < 16 && k < nAct; ++k) prefetchRecord(k);
    for (size_t k = 0; k < 8 && k < nAct; ++k) prefetchCells(k);
    for (size_t k = 0; m->deepPrefetch && k < 4 && k < nAct; ++k) prefetchEntries(k, false);
    for (size_t k = 0; m->deepPrefetch && k < 2 && k < nAct; ++k) prefetchEntries(k, true);
    size_t actIdx = 0;
    m->firstPart = -1;
    m->partEnd.clear();
    const size_t splitAt = (m->onFirstPart && nAct >= m->streamAbove && !thin) ? (size_t)((double)nAct * m->streamSplit) : (size_t)-1;
    // (sharded path: streamParts parts of equal numbers of parents)
    const int nParts = (m->onPart && nAct >= m->streamAbove && !thin) ? std::max(2, std::min(m->streamParts, PAIS_MAX_STREAM_PARTS)) : 1;
    // split points: the first part holds streamFirst of the parents (its listing is the part of the round's host work that no
    // GPU work can cover, so it is the smallest), the others share the rest equally
    auto splitPoint = [&](int k) -> size_t { // parents listed before part k begins, k = 1 .. nParts - 1
        const double f = m->streamFirst > 0 ? m->streamFirst : 1.0 / nParts;
        const double at = f + (1.0 - f) * (double)(k - 1) / (double)(nParts - 1);
        return std::max((size_t)1, std::min(nAct - 1, (size_t)((double)nAct * at)));
    };
    size_t nextSplit = nParts > 1 ? splitPoint(1) : (size_t)-1;
    for (Active &a : m->active) {
        if (actIdx == splitAt) {
            // streamed round: what has been listed so far goes to the GPU now (the callback copies the candidates)
            m->firstPart = (int)m->candRecs.size();
            if (m->firstPart > 0 && (*m->onFirstPart)(m->candRecs.data(), m->firstPart)) return -2;
        }
        if (actIdx == nextSplit && actIdx > 0) {
            const int done = m->partEnd.empty() ? 0 : m->partEnd.back(), now = (int)m->candRecs.size();
            const int part = (int)m->partEnd.size();
            m->partEnd.push_back(now);
            if (now > done && (*m->onPart)(part, m->candRecs.data() + done, now - done)) return -2;
            nextSplit = part + 2 >= nParts ? (size_t)-1 : std::max(actIdx + 1, splitPoint(part + 2)); // (the last part is handed out by the caller)
        }
        prefetchRecord(actIdx + 16);
        prefetchCells(actIdx + 8);
        if (m->deepPrefetch) {
            prefetchEntries(actIdx + 4, false);
            prefetchEntries(actIdx + 2, true);
        }
        ++actIdx;
        const pais_patch_result &pr = m->patches[a.id]->r;
        const int sEnd = thin ? pr.num_cam : a.slot + 1;
        for (int sl = a.slot; sl < sEnd; ++sl) {
            const int camI = pr.cam_idx[sl];
            const CellMap &map = m->cellMaps[camI];
            const int cx = (int)(pr.imgPoint[sl][0] / m->cfg.cellSize);
            const int cy = (int)(pr.imgPoint[sl][1] / m->cfg.cellSize);
            for (int j = 0; j < 4; ++j) considerCell(Unit{a.id, sl, j}, pr, camI, map, cx + dx[j], cy + dy[j]);
        }
        a.slot = sEnd - 1; // round_commit advances past it
    }
    if (m->truncatedVisible > 0) {
        m->truncatedVisible = 0;
        return mfail("a candidate's visibility cone holds more than PAIS_MAX_VIS cameras (patch.cpp:723-761 keeps them all): "
                     "this rig needs a larger PAIS_MAX_VIS or a larger visibleCorrelation");
    }
    *cands = m->candRecs.data();
    *n = (int)m->candRecs.size();
    m->lastEnumerateMs = now_ms() - t0;
    m->st.host_enumerate_ms += m->lastEnumerateMs;
    return 0;
}

// the sequential replay of candidates [q0, q1) of the round; results[0] is the record of candidate q0
static void commit_records(pais_mvs *m, const pais_patch_result *results, int q0, int q1)
{
    for (int q = q0; q < q1; ++q) {
        const Candidate &c = m->cands[q];
        const pais_patch_result &pr = m->patches[c.u.id]->r; // stable: patches are heap objects
        if (m->skipNeighborCell(m->cellMaps[c.cam], c.cx, c.cy, pr, -1)) continue; // mvs.cpp:558 on the live state
        m->st.candidates_effective++;
        m->st.pso_evals_effective += results[q - q0].pso_evals;
        m->insertPatch(results[q - q0]); // expandCell, mvs.cpp:576
    }
}
static int commit_finish(pais_mvs *m, int n);

extern "C" int pais_mvs_round_commit(pais_mvs *m, const pais_patch_result *results, int n)
{
    if (!m || n != (int)m->cands.size() || (n && !results)) return mfail("pais_mvs_round_commit: bad argument");
    double t0 = now_ms();
    commit_records(m, results, 0, n);
    const int rc = commit_finish(m, n);
    m->st.host_commit_ms += now_ms() - t0;
    return rc;
}

// end of a round: the cursors, the deferred units, the counters
static int commit_finish(pais_mvs *m, int n)
{
    // advance the cursors; parents that have shown all their cameras leave the set
    size_t w = 0;
    for (size_t a = 0; a < m->active.size(); ++a) {
        Active e = m->active[a];
        e.slot++;
        if (e.slot < m->patches[e.id]->r.num_cam) m->active[w++] = e;
    }
    m->active.resize(w);
    m->deferred.swap(m->nextDeferred);
    m->nextDeferred.clear();
    m->st.candidates_refined += n;
    m->st.rounds++;
    m->curRound++;
    m->cands.clear();
    return 0;
}

// ---------------------------------------------------------------- post filters ---
// The step after the path (`-f` verb, TMVS.cpp:124-172): MVS::cellFiltering, visibilityFiltering,
// neighborCellFiltering (mvs.cpp:278-446) are sequential host passes over the cell maps whose decisions depend on
// the deletions made so far -- replayed here in the reference's order; neighborPatchFiltering (:448-524) counts
// neighbours over all pairs of patches, which is the GPU kernel k_neighbor_count.

// Loader constructor Patch(center, normalS, camIdx, fitness, correlation, id), patch.cpp:45-59 -- what
// FileLoader::loadMvsPatch builds (fileloader.cpp:206-231).  Fills what the filter verbs and the writers read
// (centre, normal, cameras, fitness, correlation, image points); the fields only refine() consumes (reference
// camera, depth range, LOD, priority) are not derived on the host.
extern "C" int pais_mvs_load_patch(pais_mvs *m, const double center[3], const double normalS[2], int num_cam,
                                   const int32_t *cam_idx, double fitness, double correlation)
{
    if (!m || !center || !normalS || num_cam < 0 || num_cam > PAIS_MAX_VIS || (num_cam && !cam_idx))
        return mfail("pais_mvs_load_patch: bad argument");
    pais_patch_result r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < 3; ++i) r.center[i] = center[i];
    r.normalS[0] = normalS[0];
    r.normalS[1] = normalS[1];
    pais::spherical2normal(normalS[0], normalS[1], r.normal); // AbstractPatch::setNormal(Vec2d), abstractpatch.cpp:48-51
    r.num_cam = num_cam;
    for (int i = 0; i < num_cam; ++i) {
        if (cam_idx[i] < 0 || cam_idx[i] >= (int)m->cams.size()) return mfail("pais_mvs_load_patch: bad camera index");
        r.cam_idx[i] = cam_idx[i];
        m->project0(cam_idx[i], center, r.imgPoint[i]); // setImagePoint, patch.cpp:627-653
    }
    r.type = PAIS_TYPE_SEED;
    r.fitness = fitness;
    r.correlation = correlation;
    // setReferenceCameraIndex (patch.cpp:415-445) as the loader constructor runs it: first maximum of normal . (-optN)
    r.ref_cam = -1;
    {
        double maxCorr = -DBL_MAX;
        for (int i = 0; i < num_cam; ++i) {
            const HostCamera &cam = m->cams[cam_idx[i]];
            const double corr = r.normal[0] * (-cam.optN[0]) + r.normal[1] * (-cam.optN[1]) + r.normal[2] * (-cam.optN[2]);
            if (corr > maxCorr) { maxCorr = corr; r.ref_cam = cam_idx[i]; }
        }
    }
    r.lod = -1;
    r.key = (uint64_t)m->patches.size();
    const int id = m->storePatch(r);
    m->patches[id]->expanded = true;
    return id;
}

static void filter_prepare(pais_mvs *m)
{
    if (m->cellMaps.empty()) { // `if (cellMaps.empty()) { setNeighborRadius(); setCellMaps(); }` at the head of every filter
        m->setNeighborRadius();
        m->setCellMaps();
    }
}
// ids of a cell in the reference's vector order (insertion order; the linked list is newest-first)
static void cell_ids(const pais_mvs *m, const CellMap &map, int x, int y, std::vector<int> &out)
{
    out.clear();
    for (int e = map.first(x, y); e >= 0; e = m->pool[e].next) out.push_back(m->pool[e].id);
    std::reverse(out.begin(), out.end());
}

// MVS::cellFiltering, mvs.cpp:278-325
extern "C" int pais_mvs_cell_filtering(pais_mvs *m)
{
    if (!m) return mfail("pais_mvs_cell_filtering: bad argument");
    filter_prepare(m);
    std::vector<int> cell, removeIdx;
    for (size_t ci = 0; ci < m->cellMaps.size(); ++ci) {
        CellMap &map = m->cellMaps[ci];
        for (int x = 0; x < map.width; ++x) {
            for (int y = 0; y < map.height; ++y) {
                if (map.tileEmpty(x, y)) { y |= CellMap::kTile - 1; continue; } // same visiting order, empty tiles skipped
                if (map.first(x, y) < 0) continue;
                cell_ids(m, map, x, y, cell);
                removeIdx.clear();
                const int pthNum = (int)cell.size();
                for (int j = 0; j < pthNum; ++j) {
                    double corrSum = 0;
                    for (int k = 0; k < pthNum; ++k) {
                        if (j == k) continue;
                        const HostPatch *q = m->patches[cell[k]];
                        if (!q) continue;
                        corrSum += q->r.correlation;
                    }
                    const HostPatch *p = m->patches[cell[j]];
                    if (!p) continue;
                    if (p->r.correlation * p->r.num_cam < corrSum) removeIdx.push_back(cell[j]);
                }
                for (int id : removeIdx) m->deletePatch(id);
            }
        }
    }
    return 0;
}

// MVS::visibilityFiltering, mvs.cpp:394-446
extern "C" int pais_mvs_visibility_filtering(pais_mvs *m)
{
    if (!m) return mfail("pais_mvs_visibility_filtering: bad argument");
    filter_prepare(m);
    for (size_t id = 0; id < m->patches.size(); ++id) { // map<int,Patch> iterates in id order
        const HostPatch *p = m->patches[id];
        if (!p) continue;
        const pais_patch_result &r = p->r;
        int visibleCount = r.num_cam;
        for (int i = 0; i < r.num_cam; ++i) {
            const HostCamera &cam = m->cams[r.cam_idx[i]];
            const double d[3] = {r.center[0] - cam.C[0], r.center[1] - cam.C[1], r.center[2] - cam.C[2]};
            const double depth = pais::norm3(d);
            const int cx = (int)(r.imgPoint[i][0] / m->cfg.cellSize), cy = (int)(r.imgPoint[i][1] / m->cfg.cellSize);
            const CellMap &map = m->cellMaps[r.cam_idx[i]];
            if (!map.inMap(cx, cy)) continue; // getCell on an outside cell is undefined in the reference; loaded patches project inside
            for (int e = map.first(cx, cy); e >= 0; e = m->pool[e].next) {
                if (m->pool[e].id == (int)id) continue;
                const HostPatch *q = m->patches[m->pool[e].id];
                if (!q) continue;
                const double dn[3] = {q->r.center[0] - cam.C[0], q->r.center[1] - cam.C[1], q->r.center[2] - cam.C[2]};
                if (depth > pais::norm3(dn)) { // some other patch of the cell is in front: one occlusion per camera
                    --visibleCount;
                    break;
                }
            }
        }
        if (visibleCount < m->cfg.minCamNum) m->deletePatch((int)id);
    }
    return 0;
}

// MVS::neighborCellFiltering, mvs.cpp:327-392
extern "C" int pais_mvs_neighbor_cell_filtering(pais_mvs *m, double neighbor_ratio)
{
    if (!m) return mfail("pais_mvs_neighbor_cell_filtering: bad argument");
    filter_prepare(m);
    std::vector<int> cell, removeIdx;
    for (size_t ci = 0; ci < m->cellMaps.size(); ++ci) {
        CellMap &map = m->cellMaps[ci];
        for (int x = 0; x < map.width; ++x) {
            for (int y = 0; y < map.height; ++y) {
                if (map.tileEmpty(x, y)) { y |= CellMap::kTile - 1; continue; } // same visiting order, empty tiles skipped
                if (map.first(x, y) < 0) continue;
                cell_ids(m, map, x, y, cell);
                removeIdx.clear();
                const int nx[9] = {x, x - 1, x + 1, x - 1, x + 1, x + 1, x, x - 1, x};
                const int ny[9] = {y, y - 1, y - 1, y + 1, y + 1, y, y + 1, y, y - 1};
                for (int id : cell) {
                    const HostPatch *c = m->patches[id];
                    if (!c) continue;
                    int neighborPthSum = 0, neighborPthNum = 0;
                    for (int j = 0; j < 9; ++j) {
                        if (!map.inMap(nx[j], ny[j])) continue;
                        for (int e = map.first(nx[j], ny[j]); e >= 0; e = m->pool[e].next) {
                            ++neighborPthSum;
                            const HostPatch *q = m->patches[m->pool[e].id];
                            if (!q) continue;
                            if (m->isNeighbor(c->r, q->r)) ++neighborPthNum;
                        }
                    }
                    if ((double)neighborPthNum / (double)neighborPthSum < neighbor_ratio) removeIdx.push_back(id);
                }
                for (int id : removeIdx) m->deletePatch(id);
            }
        }
    }
    return 0;
}

// MVS::neighborPatchFiltering, mvs.cpp:448-524 (PCMVS filter): the pair counts come from the GPU
extern "C" int pais_mvs_neighbor_patch_filtering(pais_mvs *m, double neighbor_ratio, double *kernel_ms)
{
    if (!m) return mfail("pais_mvs_neighbor_patch_filtering: bad argument");
    if (!m->ctx) return mfail("pais_mvs_neighbor_patch_filtering: no GPU context (the pair counts are a HIP kernel; there is no host path)");
    filter_prepare(m);
    std::vector<int> ids;
    std::vector<double> centers;
    for (const HostPatch *p : m->patches) {
        if (!p) continue;
        ids.push_back(p->id);
        centers.insert(centers.end(), p->r.center, p->r.center + 3);
    }
    const int n = (int)ids.size();
    if (n == 0) return 0;
    std::vector<int32_t> counts((size_t)n);
    if (pais_neighbor_count(m->ctx, n, centers.data(), m->neighborRadius, counts.data(), kernel_ms))
        return mfail(pais_last_error());
    double avg = 0;
    for (int i = 0; i < n; ++i) avg += (double)counts[i];
    avg /= (double)n;
    for (int i = 0; i < n; ++i)
        if ((double)counts[i] < (avg * neighbor_ratio)) m->deletePatch(ids[i]);
    return 0;
}

extern "C" int pais_mvs_set_thin_front(pais_mvs *m, int thin_front)
{
    if (!m) return mfail("pais_mvs_set_thin_front: bad argument");
    m->thinFront = thin_front < 0 ? 0 : thin_front;
    return 0;
}

extern "C" int pais_mvs_expansion_end(pais_mvs *m)
{
    if (!m) return mfail("bad argument");
    m->setNeighborRadius(); // mvs.cpp:274
    return 0;
}

extern "C" int pais_mvs_expansion_patches(pais_mvs *m, int B, int max_rounds)
{
    if (!m || (!m->ctx && !m->recordSource)) return mfail("pais_mvs_expansion_patches: no GPU context");
    int rc = pais_mvs_expansion_begin(m);
    if (rc) return rc;
    int rounds = 0;
    for (;;) {
        const pais_candidate *c;
        const pais_patch_result *recs = m->results.data();
        int n;
        // large round: streamed (see pais_mvs::lane1) -- on one GPU, and (round 4) on the sharded path: each part is a sharded
        // batch of its own, the two exchanges enqueued on the driver's stream in part order (shard_submit / shard_finish)
        const bool oneGpu = m->ctx && m->world <= 1 && !m->nccl && !m->gatherCb && m->emuMode != 2;
        // (over a real communicator by default: yes while the canary has not failed -- the first streamed round IS the canary)
        const bool canaryRule = m->streamSharded < 0 && (m->streamCanaryMode == 1 || (m->streamCanaryMode < 0 && m->emuMode != 2));
        const bool shardStream = sharded_transport(m) && (m->streamSharded == 1 || (m->streamSharded < 0 && m->emuMode == 2 && !canaryRule) ||
                                                          (canaryRule && m->streamCanary >= 0));
        const bool canaryRound = shardStream && canaryRule && m->streamCanary == 0;
        const bool timingSaysStream = m->streamRounds == 1 && m->prevHostMs >= m->streamHostMs && m->prevHostMs >= m->streamHostShare * m->prevGpuMs;
        m->streamWish = timingSaysStream; // (travels in this rank's header with the next sharded batch)
        const bool canStream = (oneGpu || shardStream) && m->emuMode != 1 &&
                               (m->streamRounds >= 2 || (shardStream ? (m->streamRounds == 1 && m->streamAgreed) : timingSaysStream));
        int beginRc = 0;
        double tFirst = 0;
        double enqueueMs = 0; // host time of the first part's enqueue: inside round_begin, but not enumeration
        // a part of a streamed round on the sharded path: a sharded batch (device path + exchange), or -- too thin to split --
        // replicated on its lane as a host batch
        bool laneForkFailed = false; // a part (>= 1) of this round has no lane of its own
        auto part_submit = [&](int idx, pais_ctx *lane, int np, const pais_candidate *cc, int nRound) -> int {
            pais_mvs::ShardXfer &X = m->sx[idx];
            // (a part without a lane of its own must not be refined on the driver's context while part 0's asynchronous batch is
            //  pending there -- the context's per-batch state is single -- : it enters its collective with a failure status, every
            //  rank fails the round together; a replicated part fails locally, there is no collective to keep)
            const bool noLane = idx > 0 && lane == m->ctx && laneForkFailed;
            if ((long)np * m->cfg.particleNum >= (long)m->replicateBelow) {
                m->st.batches_sharded++;
                (void)pais_ctx_set_round_hint(lane, (nRound + m->world - 1) / m->world);
                return shard_submit(m, X, m->sb[idx], lane, np, cc, 0, noLane ? 1 : 0);
            }
            if (noLane) { X.open = false; return mfail("streamed sharded round: no lane for a part"); }
            m->st.batches_replicated++;
            X.sharded = false; X.lane = lane; X.n = np; X.open = true;
            (void)pais_ctx_set_round_hint(lane, nRound);
            const int rc = pais_refine_batch_begin(lane, np, cc);
            if (rc) { g_mvs_err = pais_last_error(); X.open = false; }
            return rc;
        };
        auto part_finish = [&](int idx, pais_patch_result *out, const pais_patch_result **view) -> int {
            pais_mvs::ShardXfer &X = m->sx[idx];
            if (!X.open) return 0;
            if (X.sharded) { *view = out; return shard_finish(m, X, out); }
            X.open = false;
            const int rc = pais_refine_batch_end(X.lane, view);
            if (rc) g_mvs_err = pais_last_error();
            return rc;
        };
        const std::function<int(const pais_candidate *, int)> firstPart = [&](const pais_candidate *cc, int n0) {
            tFirst = now_ms();
            (void)pais_ctx_set_round_hint(m->ctx, (int)((double)n0 / m->streamSplit)); // the round's size, as far as it is known
            beginRc = pais_refine_batch_open(m->ctx, n0, cc, m->streamHead); // the head of its launch chain; the rest below
            if (beginRc) g_mvs_err = pais_last_error();
            enqueueMs = now_ms() - tFirst;
            return beginRc;
        };
        // sharded path: part p of the round on lane p (lane 0 = the driver's own context)
        auto part_lane = [&](int p) -> pais_ctx * {
            if (p == 0) return m->ctx;
            while ((int)m->partLanes.size() < p) {
                pais_ctx *l = nullptr;
                // (a rank that cannot fork a lane must still take part in the part's collective: part_submit enters it with a
                //  failure status on the driver's own stream -- ADVICE r5: refining there would clobber part 0's pending batch)
                if (pais_ctx_fork_lane(m->ctx, &l)) { g_mvs_err = pais_last_error(); laneForkFailed = true; return m->ctx; }
                m->partLanes.push_back(l);
            }
            return m->partLanes[p - 1];
        };
        int partsOut = 0; // parts handed out so far
        const std::function<int(int, const pais_candidate *, int)> onPart = [&](int part, const pais_candidate *cc, int np) {
            const double t = now_ms();
            if (partsOut == 0) tFirst = t;
            pais_ctx *lane = part_lane(part);
            // (the round's size as far as it is known: parts of equal numbers of parents)
            beginRc = lane ? part_submit(part, lane, np, cc, np * std::max(2, std::min(m->streamParts, PAIS_MAX_STREAM_PARTS))) : -2;
            partsOut = part + 1;
            enqueueMs += now_ms() - t;
            return beginRc;
        };
        m->onFirstPart = (canStream && !shardStream) ? &firstPart : nullptr;
        m->onPart = (canStream && shardStream) ? &onPart : nullptr;
        rc = pais_mvs_round_begin(m, B, &c, &n);
        m->onFirstPart = nullptr;
        m->onPart = nullptr;
        m->lastEnumerateMs -= enqueueMs;
        m->st.host_enumerate_ms -= enqueueMs;
        // (every part in flight is waited for, in order; records of parts that cannot be used any more are dropped)
        auto drain_parts = [&](int from, int to) {
            const pais_patch_result *dummy;
            m->results.resize((size_t)std::max(std::max(n, 1), (int)m->candRecs.size()));
            for (int p = from; p < to; ++p) (void)part_finish(p, m->results.data(), &dummy);
        };
        if (rc < 0) {
            const pais_patch_result *dummy;
            if (shardStream) drain_parts(0, partsOut);
            else if (m->firstPart > 0 && !beginRc) (void)pais_refine_batch_end(m->ctx, &dummy); // nothing stays open behind an error
            m->firstPart = -1;
            return beginRc ? beginRc : rc;
        }
        if (rc == 1) { if (shardStream) drain_parts(0, partsOut); break; }
        if (shardStream && !m->partEnd.empty()) {
            // streamed round on the sharded path: parts 0 .. P-2 are on the GPU already, the rest of the list is the last part
            const double enumMs = m->lastEnumerateMs;
            int kmax = 1;
            for (int i = 0; i < n; ++i) kmax = std::max(kmax, c[i].num_cam);
            m->results.resize((size_t)std::max(n, 1));
            std::vector<int> ends(m->partEnd);
            ends.push_back(n);
            const int P = (int)ends.size();
            for (int p = 0; p < P; ++p) m->sx[p].c = c + (p ? ends[p - 1] : 0); // (the candidate list may have moved since the part was handed out)
            {
                const int lo = ends[P - 2], np = n - lo;
                if (np > 0) {
                    if (partsOut == 0) tFirst = now_ms();
                    pais_ctx *lane = part_lane(P - 1);
                    if (!lane || part_submit(P - 1, lane, np, c + lo, n)) { drain_parts(0, partsOut); return -2; }
                    partsOut = P;
                }
            }
            double commitMs = 0;
            if (canaryRound && n > 0) {
                // the canary: every part finished first (nothing committed), then the round once more as ONE sharded batch; the two
                // record sets must be the same bytes on every rank, or streamed sharded rounds stay off for this driver
                m->canaryRecs.resize((size_t)n);
                for (int p = 0; p < P; ++p) {
                    const int lo = p ? ends[p - 1] : 0, np = ends[p] - lo;
                    if (np <= 0) continue;
                    const pais_patch_result *v = nullptr;
                    if (part_finish(p, m->canaryRecs.data() + lo, &v)) { drain_parts(p + 1, P); return -2; }
                    if (v != m->canaryRecs.data() + lo) memcpy(m->canaryRecs.data() + lo, v, sizeof(pais_patch_result) * (size_t)np);
                }
                int rc2 = shard_submit(m, m->sx[0], m->sb[0], m->ctx, n, c, 0);
                if (!rc2) rc2 = shard_finish(m, m->sx[0], m->results.data());
                if (rc2) return rc2;
                const bool same = memcmp(m->canaryRecs.data(), m->results.data(), sizeof(pais_patch_result) * (size_t)n) == 0;
                const int verdict = shard_growth_handshake(m, same ? 0 : -7); // (4 bytes per rank: every rank takes the same decision)
                m->streamCanary = verdict ? -1 : 1;
                if (getenv("PAIS_STREAM_CANARY_LOG"))
                    fprintf(stderr, "[pais] rank %d: streamed sharded round of %d candidates in %d parts %s its unstreamed replay -> streaming %s\n",
                            m->rank, n, P, same ? "==" : "!=", m->streamCanary > 0 ? "on" : "off");
                const double t1 = now_ms();
                commit_records(m, m->results.data(), 0, n);   // (the unstreamed records: the path every test has verified)
                commitMs += now_ms() - t1;
            } else
            for (int p = 0; p < P; ++p) {
                const int lo = p ? ends[p - 1] : 0, np = ends[p] - lo;
                if (np <= 0) continue;
                const pais_patch_result *v = nullptr;
                if (part_finish(p, m->results.data() + lo, &v)) { drain_parts(p + 1, P); return -2; }
                const double t1 = now_ms();
                commit_records(m, v, lo, ends[p]);
                commitMs += now_ms() - t1;
            }
            const double t2 = now_ms();
            rc = commit_finish(m, n);
            commitMs += now_ms() - t2;
            m->st.host_commit_ms += commitMs;
            if (rc) return rc;
            const double tRef = n > 0 ? std::max(0.0, (now_ms() - tFirst) - commitMs) : 0.0;
            m->st.gpu_refine_ms += tRef;
            m->st.rounds_streamed++;
            if (n > 0) m->roundLog.push_back(pais_round_log{n, 0, 1, kmax, tRef, enumMs, commitMs});
            m->prevHostMs = enumMs + commitMs;
            m->prevGpuMs = tRef + commitMs;
            if (max_rounds > 0 && ++rounds >= max_rounds) break;
            continue;
        }
        if (m->firstPart >= 0) {
            const int n0 = m->firstPart, n1 = n - n0;
            const double enumMs = m->lastEnumerateMs;
            int kmax = 1;
            for (int i = 0; i < n; ++i) kmax = std::max(kmax, c[i].num_cam);
            if (n1 > 0) {
                if (!m->lane1 && pais_ctx_fork_lane(m->ctx, &m->lane1)) { g_mvs_err = pais_last_error(); return -2; }
                if (n0 == 0) tFirst = now_ms();
                (void)pais_ctx_set_round_hint(m->lane1, n);
                if (pais_refine_batch_open(m->lane1, n1, c + n0, m->streamHead)) {
                    g_mvs_err = pais_last_error();
                    const pais_patch_result *dummy;
                    if (n0 > 0) (void)pais_refine_batch_end(m->ctx, &dummy); // nothing stays open behind an error
                    return -2;
                }
            }
            // the two launch chains, a few PSO iterations of each in turn: neither lane waits for the other's whole chain
            for (int a = n0 > 0 ? 0 : 1, b = n1 > 0 ? 0 : 1; !a || !b;) {
                if (!a) a = pais_refine_batch_enqueue(m->ctx, m->streamStep);
                if (!b && a >= 0) b = pais_refine_batch_enqueue(m->lane1, m->streamStep);
                if (a < 0 || b < 0) {
                    g_mvs_err = pais_last_error();
                    const pais_patch_result *dummy;
                    if (a >= 0 && n0 > 0) (void)pais_refine_batch_end(m->ctx, &dummy);
                    if (b >= 0 && n1 > 0) (void)pais_refine_batch_end(m->lane1, &dummy);
                    return -2;
                }
            }
            double commitMs = 0;
            const pais_patch_result *v0 = nullptr, *v1 = nullptr;
            if (n0 > 0) {
                if (pais_refine_batch_end(m->ctx, &v0)) {
                    g_mvs_err = pais_last_error();
                    if (n1 > 0) (void)pais_refine_batch_end(m->lane1, &v1);
                    return -2;
                }
                const double t1 = now_ms();
                commit_records(m, v0, 0, n0);
                commitMs += now_ms() - t1;
            }
            if (n1 > 0 && pais_refine_batch_end(m->lane1, &v1)) { g_mvs_err = pais_last_error(); return -2; }
            const double t2 = now_ms();
            if (n1 > 0) commit_records(m, v1, n0, n);
            rc = commit_finish(m, n);
            commitMs += now_ms() - t2;
            m->st.host_commit_ms += commitMs;
            if (rc) return rc;
            // GPU time that the host waited for: from the first launch to the end of the round, less the commits inside it
            const double tRef = n > 0 ? std::max(0.0, (now_ms() - tFirst) - commitMs) : 0.0;
            m->st.gpu_refine_ms += tRef;
            m->st.rounds_streamed++;
            if (n > 0) m->roundLog.push_back(pais_round_log{n, 0, 0, kmax, tRef, enumMs, commitMs});
            m->prevHostMs = enumMs + commitMs;
            m->prevGpuMs = tRef + commitMs; // (what the GPU was busy for, roughly: the wait and the commit inside it)
            if (max_rounds > 0 && ++rounds >= max_rounds) break;
            continue;
        }
        m->results.resize((size_t)(n > 0 ? n : 1));
        const double enumMs = m->lastEnumerateMs;
        double tRef = 0;
        const int64_t shardedBefore = m->st.batches_sharded;
        int kmax = 1;
        if (n > 0) {
            for (int i = 0; i < n; ++i) kmax = std::max(kmax, c[i].num_cam);
            double t0 = now_ms();
            rc = refine_any(m, n, c, m->results.data(), 0, &recs);
            tRef = now_ms() - t0;
            m->st.gpu_refine_ms += tRef;
            if (rc) return rc;
        }
        const double t1 = now_ms();
        rc = pais_mvs_round_commit(m, n > 0 ? recs : m->results.data(), n);
        if (rc) return rc;
        const double commitMs1 = now_ms() - t1;
        if (n > 0) m->roundLog.push_back(pais_round_log{n, 0, m->st.batches_sharded > shardedBefore ? 1 : 0, kmax, tRef, enumMs, commitMs1});
        m->prevHostMs = enumMs + commitMs1;
        m->prevGpuMs = tRef;
        if (max_rounds > 0 && ++rounds >= max_rounds) break;
    }
    return pais_mvs_expansion_end(m);
}

// ------------------------------------------------------ seeds from features ---
// FeatureManager::setSeedPatches after the descriptor matching (mvs/featuremanager.cpp:41-99, 118-243); include/pais_seed.h.
// The reference keeps vectors of DMatch per ordered camera pair and erases from them while it scans; the outcome of those
// scans is stated here over hashed pair sets and a (camera, feature) -> first-feature index, which give the same lists.
extern "C" int pais_mvs_seeds_from_matches(pais_mvs *m, int num_cams, const pais_keypoints *kp, int num_matches,
                                           const pais_pair_match *matches, double max_dist, int *num_seeds)
{
    if (!m || !kp || num_cams != (int)m->cams.size() || num_matches < 0 || (num_matches && !matches))
        return mfail("pais_mvs_seeds_from_matches: bad argument");
    const int C = num_cams;
    for (int c = 0; c < C; ++c)
        if (kp[c].n < 0 || (kp[c].n && !kp[c].xy)) return mfail("pais_mvs_seeds_from_matches: bad keypoints");
    // fundamental matrices, getFundamentalMatrices (:265-287): M[i][j] = F(from j, to i) for i < j, transposed below it
    std::vector<double> Fs((size_t)C * C * 9, 0.0);
    {
        std::vector<pais_camera_desc> d((size_t)C);
        for (int c = 0; c < C; ++c) {
            memset(&d[c], 0, sizeof(pais_camera_desc));
            memcpy(d[c].KR, m->cams[c].KR, sizeof(d[c].KR));
            memcpy(d[c].KT, m->cams[c].KT, sizeof(d[c].KT));
            memcpy(d[c].center, m->cams[c].C, sizeof(d[c].center));
        }
        for (int i = 0; i < C; ++i)
            for (int j = i; j < C; ++j) {
                double *Fij = &Fs[((size_t)i * C + j) * 9], *Fji = &Fs[((size_t)j * C + i) * 9];
                if (i == j) { Fij[0] = Fij[4] = Fij[8] = 1.0; continue; }
                if (pais_seed_fundamental(&d[j], &d[i], Fij)) return mfail(pais_seed_last_error());
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) Fji[r * 3 + c] = Fij[c * 3 + r];
            }
    }
    // the match table after epipolarLineFiltering (:158-196)
    typedef std::pair<int, int> QT;
    std::vector<std::vector<QT>> table((size_t)C * C);
    {   // a tuple given twice is refused (the statement below over hashed pair sets equals the reference's
        // erase-while-scanning lists only while the matches of a pair are unique)
        std::unordered_set<uint64_t> seen;
        seen.reserve((size_t)num_matches * 2);
        for (int k = 0; k < num_matches; ++k) {
            const pais_pair_match &mm = matches[k];
            if (mm.cam_q < 0 || mm.cam_q >= C || mm.cam_t < 0 || mm.cam_t >= C || mm.q < 0 || mm.t < 0 || mm.q >= (1 << 24) || mm.t >= (1 << 24) || C > 256)
                continue; // (validated below / rigs beyond the key's range: checked per pair list instead)
            const uint64_t kk = ((uint64_t)mm.cam_q << 56) | ((uint64_t)mm.cam_t << 48) | ((uint64_t)mm.q << 24) | (uint64_t)mm.t;
            if (!seen.insert(kk).second) return mfail("pais_mvs_seeds_from_matches: the same (cam_q, cam_t, q, t) match is given twice");
        }
    }
    for (int k = 0; k < num_matches; ++k) {
        const pais_pair_match &mm = matches[k];
        if (mm.cam_q < 0 || mm.cam_q >= C || mm.cam_t < 0 || mm.cam_t >= C || mm.cam_q == mm.cam_t || mm.q < 0 || mm.q >= kp[mm.cam_q].n ||
            mm.t < 0 || mm.t >= kp[mm.cam_t].n)
            return mfail("pais_mvs_seeds_from_matches: bad match");
        const double *F = &Fs[((size_t)mm.cam_q * C + mm.cam_t) * 9];
        const double qx = kp[mm.cam_q].xy[2 * mm.q], qy = kp[mm.cam_q].xy[2 * mm.q + 1];
        const double tx = kp[mm.cam_t].xy[2 * mm.t], ty = kp[mm.cam_t].xy[2 * mm.t + 1];
        double l[3];
        for (int c = 0; c < 3; ++c) l[c] = qx * F[c] + qy * F[3 + c] + 1.0 * F[6 + c]; // q^T F
        const double dist = fabs(l[0] * tx + l[1] * ty + l[2] * 1.0) / sqrt(l[0] * l[0] + l[1] * l[1]);
        if (dist > max_dist) continue;
        table[(size_t)mm.cam_q * C + mm.cam_t].push_back(QT(mm.q, mm.t));
    }
    // filteroutNonMatches (:198-243), first half: a match of (i, j) stays iff (j, i) still holds its mirror image, which is
    // consumed.  Matches of a pair are unique per query, so "the first mirror found" is "the mirror".
    auto key = [](int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; };
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            std::vector<QT> &a = table[(size_t)i * C + j], &b = table[(size_t)j * C + i];
            if (a.empty()) continue;
            std::unordered_set<uint64_t> mirror;
            for (const QT &x : b) mirror.insert(key(x.second, x.first)); // as (q of a, t of a)
            std::unordered_set<uint64_t> consumed;
            std::vector<QT> keep;
            for (const QT &x : a)
                if (mirror.count(key(x.first, x.second)) && !consumed.count(key(x.first, x.second))) {
                    keep.push_back(x);
                    consumed.insert(key(x.first, x.second));
                }
            if (i != j) {
                std::vector<QT> rest;
                for (const QT &x : b)
                    if (!consumed.count(key(x.second, x.first))) rest.push_back(x);
                b.swap(rest);
            }
            a.swap(keep);
        }
    // ... second half: views with fewer than a quarter of the camera's best match count are dropped
    for (int i = 0; i < C; ++i) {
        size_t maxMatch = 0;
        for (int j = 0; j < C; ++j) maxMatch = std::max(maxMatch, table[(size_t)i * C + j].size());
        for (int j = 0; j < C; ++j)
            if ((double)table[(size_t)i * C + j].size() < (double)maxMatch / 4.0) table[(size_t)i * C + j].clear();
    }
    // union (:56-82, setNVMatch :118-156): a match joins the FIRST n-view feature that holds one of its two ends -- within
    // that feature the first element that is one of them decides which end is the link -- else it opens a feature
    struct Node { int cam, feat; };
    std::vector<std::vector<Node>> nv;
    std::unordered_map<uint64_t, int> firstFeature; // (cam, feat) -> lowest index of a feature holding it
    auto hold = [&](int u, int cam, int feat) {
        auto it = firstFeature.find(key(cam, feat));
        if (it == firstFeature.end() || it->second > u) firstFeature[key(cam, feat)] = u;
    };
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
            if (i == j) continue;
            std::vector<QT> &l = table[(size_t)i * C + j];
            while (!l.empty()) {
                const QT mt = l.back();
                int u = -1;
                auto a = firstFeature.find(key(i, mt.first)), b = firstFeature.find(key(j, mt.second));
                if (a != firstFeature.end()) u = a->second;
                if (b != firstFeature.end() && (u < 0 || b->second < u)) u = b->second;
                if (u >= 0) {
                    std::vector<Node> &f = nv[(size_t)u];
                    for (size_t e = 0; e < f.size(); ++e) {
                        const bool isQ = f[e].cam == i && f[e].feat == mt.first, isT = f[e].cam == j && f[e].feat == mt.second;
                        if (!isQ && !isT) continue;
                        const int addCam = isQ ? j : i, addFeat = isQ ? mt.second : mt.first; // (the query test comes first)
                        bool present = false;
                        for (const Node &x : f) present = present || x.feat == addFeat; // the reference compares the feature index alone
                        if (!present) {
                            f.push_back(Node{addCam, addFeat});
                            hold(u, addCam, addFeat);
                        }
                        break;
                    }
                } else {
                    nv.push_back(std::vector<Node>{Node{i, mt.first}, Node{j, mt.second}});
                    hold((int)nv.size() - 1, i, mt.first);
                    hold((int)nv.size() - 1, j, mt.second);
                }
                l.pop_back();
            }
        }
    // seeds (:84-99)
    int added = 0;
    const double zero[3] = {0, 0, 0};
    for (const std::vector<Node> &f : nv) {
        if ((int)f.size() < m->cfg.minCamNum) continue;
        if (f.size() > (size_t)PAIS_MAX_VIS) return mfail("pais_mvs_seeds_from_matches: an n-view feature holds more than PAIS_MAX_VIS views");
        int32_t cams[PAIS_MAX_VIS];
        double pts[2 * PAIS_MAX_VIS];
        for (size_t e = 0; e < f.size(); ++e) {
            cams[e] = f[e].cam;
            pts[2 * e] = (double)kp[f[e].cam].xy[2 * f[e].feat];
            pts[2 * e + 1] = (double)kp[f[e].cam].xy[2 * f[e].feat + 1];
        }
        const int id = pais_mvs_add_seed_measured(m, zero, (int)f.size(), cams, pts, 1);
        if (id < 0) return id;
        ++added;
    }
    if (num_seeds) *num_seeds = added;
    return 0;
}

extern "C" int pais_mvs_set_seed_patches(pais_mvs *m, int num_cams, const pais_keypoints *kp, int dim, double max_dist, int *num_seeds)
{
    if (!m || !kp || num_cams != (int)m->cams.size() || dim <= 0) return mfail("pais_mvs_set_seed_patches: bad argument");
    if (!m->ctx) return mfail("pais_mvs_set_seed_patches: this driver owns no GPU (the descriptor matching runs on it)");
    // nearest(i -> j) once per ordered pair, every camera's descriptors uploaded once; the cross-check of (i, j) reads the
    // two directions (BFMatcher crossCheck: q keeps its nearest t only if t's nearest is q)
    size_t total = 0;
    std::vector<size_t> pairOff((size_t)num_cams * num_cams, 0);
    for (int i = 0; i < num_cams; ++i)
        for (int j = 0; j < num_cams; ++j) {
            if (i == j) continue;
            pairOff[(size_t)i * num_cams + j] = total;
            total += (size_t)std::max(kp[i].n, 0);
        }
    std::vector<int32_t> nn(total ? total : 1, -1);
    if (pais_seed_nearest_all(m->device, num_cams, kp, dim, nn.data())) return mfail(pais_seed_last_error());
    std::vector<pais_pair_match> all;
    for (int i = 0; i < num_cams; ++i)
        for (int j = 0; j < num_cams; ++j) {
            if (i == j) continue;
            const int32_t *ij = nn.data() + pairOff[(size_t)i * num_cams + j], *ji = nn.data() + pairOff[(size_t)j * num_cams + i];
            for (int q = 0; q < kp[i].n; ++q) {
                const int32_t t = ij[q];
                if (t >= 0 && t < kp[j].n && ji[t] == q) all.push_back(pais_pair_match{i, j, q, t});
            }
        }
    return pais_mvs_seeds_from_matches(m, num_cams, kp, (int)all.size(), all.data(), max_dist, num_seeds);
}

extern "C" int pais_mvs_num_patches(const pais_mvs *m) { return m ? m->alive : 0; }
extern "C" int pais_mvs_num_slots(const pais_mvs *m) { return m ? (int)m->patches.size() : 0; }
extern "C" int pais_mvs_get_patch(const pais_mvs *m, int id, pais_patch_result *out, int *expanded)
{
    if (!m || id < 0 || id >= (int)m->patches.size() || !m->patches[id]) return 1;
    if (out) {
        memset(out, 0, sizeof(*out)); // (array tails beyond num_cam: zeros, as the batch calls leave them)
        pais_mvs::copyRecordCompact(out, m->patches[id]->r);
    }
    if (expanded) *expanded = m->patches[id]->expanded ? 1 : 0;
    return 0;
}
extern "C" double pais_mvs_neighbor_radius(const pais_mvs *m) { return m ? m->neighborRadius : 0.0; }
extern "C" int pais_mvs_get_stats(const pais_mvs *m, pais_mvs_stats *out)
{
    if (!m || !out) return -1;
    *out = m->st;
    return 0;
}
extern "C" int pais_mvs_get_round_log(const pais_mvs *m, pais_round_log *out, int cap)
{
    if (!m) return 0;
    const int n = (int)m->roundLog.size();
    for (int i = 0; out && i < n && i < cap; ++i) out[i] = m->roundLog[i];
    return n;
}
extern "C" const char *pais_mvs_last_error(void) { return g_mvs_err.c_str(); }

