// kbest_capi.cpp -- the C ABI of include/kbest_c.h on top of the HIP kernels.
// Plain pointers and sizes only; no torch types.  There is no CPU fallback:
// every compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "kbest_c.h"
#include "kbest_engine.h"

namespace kb {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

struct DevBufRaw { void *p = nullptr; size_t bytes = 0; };

// A few host threads for the host-side half of the host-buffer entry (widening byte tables into the caller's int32 tables while
// the GPU works on the next piece): ONE pool per process, shared by every context (a pool per context ran a test session with
// dozens of contexts into the host's thread limit), created at first use, parked on a condition variable between calls, never
// destroyed (the process ends with its threads parked).
struct HostPool {
    std::mutex runMu;  // one parallel job at a time
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cvGo, cvDone;
    const std::function<void(int)> *job = nullptr;
    std::atomic<int> next{0};
    int nTasks = 0, busy = 0;
    unsigned long long gen = 0;
    bool quit = false;
    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)> *j;
            int n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cvGo.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                j = job;
                n = nTasks;
            }
            for (int t; (t = next.fetch_add(1, std::memory_order_relaxed)) < n;) (*j)(t);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--busy == 0) cvDone.notify_all();
            }
        }
    }
    void start(int n)
    {
        for (int i = 0; i < n; i++) {
            try {
                th.emplace_back([this] { worker(); });
            } catch (const std::system_error &) {  // (no more threads to be had: the pool is as large as it got)
                break;
            }
        }
    }
    // fn(0 .. n-1), spread over the pool and the calling thread; returns when all are done
    void run(int n, const std::function<void(int)> &fn)
    {
        std::lock_guard<std::mutex> one(runMu);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn;
            nTasks = n;
            next.store(0, std::memory_order_relaxed);
            busy = (int)th.size();
            gen++;
        }
        cvGo.notify_all();
        for (int t; (t = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(t);
        std::unique_lock<std::mutex> lk(mu);
        cvDone.wait(lk, [&] { return busy == 0; });
    }
};

static HostPool *host_pool(int want)  // want: threads besides the caller (first call decides)
{
    static std::once_flag once;
    static HostPool *pool = nullptr;
    std::call_once(once, [&] {
        pool = new HostPool;
        pool->start(want);
    });
    return pool;
}

struct kbest_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // the other pieces of a large host-entry batch (kbest_batch_f64)
    hipStream_t hi = nullptr;                          // ... and the first piece's (highest priority)
    hipStream_t copy = nullptr;                        // uploads of registered cost blocks, piece after piece (narrow staging)
    int prioMain = 0, prioAux[3] = {0, 0, 0};          // their stream priorities (kbest_create)
    unsigned char *states = nullptr;  // hypothesis-state workspace (+ the slot -> state table behind it)
    size_t statesBytes = 0;
        unsigned char *wide = nullptr;    // work space of the general-size kernel (kbest_wide.hip)
    size_t wideBytes = 0;
    unsigned *wideQueue = nullptr;  // the general-size kernel's problem queue (two words, zero between launches)
    bool noWideQueue = false;       // KBEST_NO_WIDE_QUEUE: fixed stride over the batch (A/B)
    int ldsLimit = 65536;
    int nWaves = 0;   // waves per cost matrix (workgroup = nWaves * 64 threads); 0 = choose per launch
    int spec = 0;     // candidates re-solved / split per round; 0 = choose per launch (choose_spec)
    int ldsPerCU = 160 * 1024;
    int nCU = 256;
    int smallWaves = 0;  // waves per problem of the small-problem kernel; 0 = choose per launch (KBEST_SMALL_NW)
    int bnbSmallFrom = -1; // KBEST_BNB_SMALL_FROM: batch size from which the bounded walk runs in 256-thread workgroups (-1 = nCU + 1)
    bool noBnb = false;  // KBEST_NO_BNB: no bounded-walk kernel (kbest_bnb.hip) on the association path (A/B, tests)
    bool noTiny = false; // KBEST_NO_TINY: frames with a handful of measurements through the enumeration kernels too (A/B, tests)
    bool noT0 = false;        // KBEST_NO_T0: no a-priori threshold in the 64-row kernel (A/B tests)
    int wideNw = 0;           // KBEST_WIDE_NW: waves per problem of the general-size kernel (8 / 16; A/B tests)
    int wideTile = -1;        // KBEST_WIDE_TILE: 0 / 1 force the cost copy out of / into LDS (A/B tests)
    int wideSpec = 0;         // KBEST_WIDE_SPEC: hypotheses split per round by the general-size kernel (A/B tests)
    bool exactRoot = false;   // KBEST_EXACT_ROOT: the 64-row kernel's root without the column-reduction start (A/B tests)
    size_t zcLimit = (size_t)64 << 20;  // KBEST_ZC_LIMIT_KB: association calls up to this many bytes in + out run without device copies
    bool noPoll = false;      // KBEST_NO_POLL: zero-copy calls wait for the stream instead of polling the completion counter
    bool forceSmall = false;  // KBEST_FORCE_SMALL: every batch of <= 32-row problems through the small-problem kernel
    bool noSmall = false;  // KBEST_NO_SMALL: problems of <= 32 rows through the 64-row kernel as well (A/B tests)
    bool forceWide = false;   // KBEST_FORCE_WIDE: everything through the general-size kernel (test hook; read once at create)
    bool noSplit = false;     // KBEST_NO_SPLIT: never split one matrix over several workgroups (A/B tests)
    bool noTie = false;       // KBEST_NO_TIE: no extra solution / canonical order of exact ties (A/B tests; kbest_ties.h)
    int splitForce = 0;       // KBEST_SPLIT: workgroups per matrix (2 / 4) whenever the split is possible (A/B tests)
    DevBufRaw splitBuf;       // per-share result tables + shared thresholds of the split
    DevBufRaw tieBuf;         // [B] fp64: gain of the solution behind the tables (exact ties, kbest_ties.h)
    DevBufRaw exactBuf;       // work space of the reference-order kernel (kbest_exact.hip)
    DevBufRaw relayBuf;       // relay launches of the 64-row kernel: [B] LDS images (kbest_engine.hip)
    DevBufRaw relayFlags;     // ... and three words per matrix: claimed / done / gone (zeroed when the buffer is made, put back to zero by every launch)
    long long relayLaunches = 0;  // relay launches made (kbest_relay_launches)
    std::atomic<bool> relayDirty{false};  // an entry of this context failed (HIP error, nf < 0): the relay's words are zeroed before the next relay launch
    bool relayCaptured = false;   // a relay launch was captured into a graph: the graph holds the relay work space's addresses
    int lastRoute = 0;            // which kernel(s) the last k-best launch went to (kbest_last_route)
    int refOrder = 0;             // kbest_set_reference_order: 1 = the association entries enumerate in the reference's own order (kbest_exact.hip);
                                  // 2 = only the frames whose k-th and (k+1)-th gains are equal do (their weights are then the reference's)
    int relay = -1;           // KBEST_RELAY: pieces per matrix (0 / 1: never; -1: choose per launch)
    int relayFirst = 0;       // KBEST_RELAY_FIRST: the first piece hands over at k * this / 1024 solutions (0: choose per launch shape)
    int relayStep = 0;        // KBEST_RELAY_STEP: the later pieces hand over this / 1024 of k apart (0: even steps up to k)
    std::vector<int32_t> lastTie;  // KBEST_TIE_* per problem of the last synchronous call (kbest_last_tie_flags)
    int32_t *assocTieDev = nullptr;  // where kbest_assoc_probs_batch_f64_dev writes its flags (kbest_set_assoc_tie_flags_dev)
    std::mutex tieMu;
    bool noReorder = false;   // KBEST_NO_REORDER: the 64-row kernel enumerates in the reference's column order (A/B tests)
    int zcCost = 1;           // KBEST_ZC_COST=0: cost blocks in registered memory are copied up first instead of read in place (A/B tests)
    int pieces = 0;           // KBEST_PIECES: pieces of a large host-entry batch (1 / 2 / 4; A/B tests); 0 = choose
    bool noLane = false;      // KBEST_NO_LANE: no lane-per-child kernel (A/B tests)
    bool forceLane = false;   // KBEST_FORCE_LANE: every plain batch of <= 32-row problems through the lane-per-child kernel
    int laneNw = 0;           // KBEST_LANE_NW: waves per problem of the lane-per-child kernel (1 / 2 / 4); 0 = choose per launch
    int laneSpec = 0;         // KBEST_LANE_SPEC: hypotheses split per round there (1 .. 16); 0 = choose per launch
    // optimistic bounds of the 64-row kernel (kbest_engine.hip, struct Opt); < 0: choose per launch shape (opt_defaults)
    float optRho0 = -1.0f, optRho1 = -1.0f, optPhi = -1.0f, optKappa = 0.25f;
    int optMinPool = 8;
    bool noOpt = false;       // KBEST_NO_OPT: no optimistic bounds (A/B tests)
    int extraStates = 64;  // lazy state slots beyond k per matrix (room for speculative re-solves)
    int eagerStates = 1024; // state slots per matrix for children that are kept in full when they are found
    unsigned long long *prof = nullptr;  // diagnostic builds only (kbest_set_profile_buffer)
    std::string err;
    std::mutex errMu;  // entry points may be called from several host threads
    std::recursive_mutex mu;  // serialises launches: the hypothesis workspace is shared by every launch of this context (recursive: the
                              // host entry holds it across all pieces of a pieced launch, each of which takes it again)
    // The workspace is per context, not per stream: a launch on another stream than the previous one must wait for it
    // (an event recorded on the previous stream when the switch is detected).
    hipStream_t lastStream = nullptr;
    bool haveLast = false;
    hipEvent_t lastEvent = nullptr;
    // Pinned, device-mapped staging of the per-frame association path: the reference calls getAssignmentProbs once
    // per frame (system.cpp:268), so a call must cost one launch, not a dozen copies.  Small calls are zero-copy (the
    // kernel reads the cost block from and writes the probabilities to this host memory directly); larger ones go
    // through one asynchronous copy each way.
    struct Arena { void *host = nullptr; void *dev = nullptr; size_t bytes = 0; };
    Arena pinIn, pinOut;
    Arena pinTab;             // byte tables + gains + counts of the host-buffer entry's narrow staging (kbest_batch_f64)
    std::mutex narrowMu;      // one narrow-staged call at a time per context (the staging memory and the pool are the context's)
    int hostThreads = 0;      // KBEST_HOST_THREADS: size of the process's pool of widening threads when this context creates it (0 = choose: up to 15 + the caller)
    bool noNarrow = false;    // KBEST_NO_NARROW: int32 tables cross PCIe as they are (A/B tests)
    DevBufRaw stageIn, stageOut;
    // Device buffers of the host-pointer entry points are recycled: the reference calls assignmentProb once per
    // frame, and a dozen hipMalloc/hipFree pairs per call cost more than the kernels of a 30 x 10 problem.
    // Host memory the caller has registered (kbest_register_host_buffer): pinned and mapped into the device's address space,
    // so that result tables go there straight from the kernel and cost blocks come up with asynchronous copies.
    struct HostReg { char *host; char *dev; size_t bytes; };
    std::vector<HostReg> regs;
    std::mutex regMu;
    struct Block { void *p; size_t n; bool used; };
    std::vector<Block> cache;
    size_t cacheBytes = 0;  // all blocks, idle or in use
    std::mutex cacheMu;
};

namespace {

int fail(kbest_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        // (a launch that failed may have left the relay's per-matrix words non-zero: kbest_engine.hip, relay_depart)
        if (code == KBEST_ERR_HIP || code == KBEST_ERR_INTERNAL) ctx->relayDirty.store(true, std::memory_order_relaxed);
        std::lock_guard<std::mutex> lock(ctx->errMu);
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                   \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail(ctx, KBEST_ERR_HIP, #call, e_);    \
    } while (0)

// Launch shape, tuned on MI355X (DESIGN.md sections 3, 8).  The 32 KiB cost tile of a 64-row problem limits a CU to three
// resident matrices and 80 VGPRs limit it to 24 waves: {8 waves, 6 candidates, 3 matrices/CU} is the throughput shape,
// {12 waves, 12 candidates, 2/CU} and {16 waves, 16 candidates, 1/CU} trade throughput for latency when the batch is small.
// (Since the a-priori thresholds bound the early rounds, speculative splits are cheap: 1 024 x 64x64 at 12 waves,
//  8 / 10 / 12 candidates per round: 2.97 / 2.80 / 2.74 ms.)
struct Shape { int nWaves, spec; int lanes = 4; };

Shape choose_shape(const kbest_ctx *ctx, int B, int maxRow, int k)
{
    Shape s;
    // A batch that cannot fill the chip is a latency problem: the reference calls assignmentProb once per frame.
    // With at most one (two) matrices per CU the whole CU (half of it) goes to each: 16 (12) waves, 8 candidates per
    // round -- 64x64, k = 200 alone: 1.25 ms instead of 1.80; one 30x10 frame: 0.63 ms instead of 0.99.
    if (B <= ctx->nCU) { s.nWaves = 16; s.spec = 16; }  // 16 candidates per round (one 30x10 frame 0.63 -> 0.57 ms; 256 x 64x64 1.10 -> 1.03 ms)
    else if (B <= 2 * ctx->nCU) { s.nWaves = 12; s.spec = 12; }
    else if (maxRow <= 32) {
        // (32x32, k = 200, ms for 4 / 8 / 12 waves with as many hypotheses per round: B = 2 048: 2.28 / 2.12 / 2.29, 4 096: 3.74 /
        //  3.86 / 4.33, 8 192: 6.81 / 7.30 / 8.39)
        if (B <= 12 * ctx->nCU) { s.nWaves = 8; s.spec = 8; } else { s.nWaves = 4; s.spec = 4; }
    }
    else {
        // measured (64x64, k = 200, ms for 8 / 12 waves): B = 600: 2.28 / 2.48, 900: 3.35 / 3.05, 1024: 4.0 / 3.2,
        // 1100: 3.57 / 3.74, 1536: 4.27 / 4.60, 2048: 5.62 / 5.98, 8192: 19.4 / 21.9 -- three 8-wave matrices per CU win
        // except where they would leave a nearly empty second generation and two generations of 12-wave pairs fit
        // (re-scanned after the enumeration got its own column order, DESIGN.md section 2 point 8 -- fewer, shorter children: the
        //  12-wave pairs now win at every batch size, ms for 8 / 12 waves: B = 600: 1.85 / 1.61, 768: 2.02 / 1.78, 1 100: 2.71 /
        //  2.44, 1 536: 3.36 / 2.98, 2 048: 4.27 / 3.84, 4 096: 7.95 / 7.26)
        s.nWaves = 12;
        s.spec = 12;
    }
    // the in-place pool merge holds at most 4 entries per thread: a long pool (bruteForceProb-style k in the
    // thousands, assignment.cpp:868) needs a bigger workgroup
    while (k > 4 * s.nWaves * 64 && s.nWaves < 16) s.nWaves = (s.nWaves < 8) ? 8 : 16;
    if (ctx->nWaves > 0) s.nWaves = ctx->nWaves;
    if (ctx->spec > 0) s.spec = ctx->spec;
    if (s.spec > s.nWaves) s.spec = s.nWaves;
    if (s.spec > 8 && s.nWaves < 12) s.spec = 8;   // selection slots of the kernel: 8, 12 from 12 waves, 16 from 16 waves on
    if (s.spec > 12 && s.nWaves < 16) s.spec = 12;
    while (s.spec > 1 && kb::lds_layout(maxRow, k, s.spec, s.nWaves).total > ctx->ldsLimit) s.spec /= 2;
    return s;
}

// NOTE: the context's stream is created non-blocking, i.e. it does NOT order itself against the legacy default
// stream.  Every device-side initialisation of the host-pointer entry points therefore goes through
// hipMemsetAsync on the context's own stream (a plain hipMemset could still be in flight when the kernel runs and
// wipe what the kernel has already written).  Host-to-device copies from pageable memory are complete on return.
struct DevBuf {  // RAII device buffer of the host-pointer entry points, drawn from the context's block cache
    void *p = nullptr;
    kbest_ctx *owner = nullptr;
    ~DevBuf()
    {
        if (!p) return;
        std::lock_guard<std::mutex> lock(owner->cacheMu);
        for (auto it = owner->cache.begin(); it != owner->cache.end(); ++it)
            if (it->p == p) {
                // one-off giants are not kept, and the cache as a whole stays below 4 GiB
                if (it->n > ((size_t)256 << 20) || owner->cacheBytes > ((size_t)4 << 30)) {
                    owner->cacheBytes -= it->n;
                    (void)hipFree(p);
                    owner->cache.erase(it);
                } else {
                    it->used = false;
                }
                return;
            }
    }
    hipError_t alloc(kbest_ctx *ctx, size_t n)
    {
        owner = ctx;
        if (n == 0) n = 1;
        std::lock_guard<std::mutex> lock(ctx->cacheMu);
        kbest_ctx::Block *best = nullptr;
        for (auto &b : ctx->cache)
            if (!b.used && b.n >= n && b.n <= 8 * n + 4096 && (!best || b.n < best->n)) best = &b;
        if (best) { best->used = true; p = best->p; return hipSuccess; }
        const size_t cap = (n + 4095) & ~(size_t)4095;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) {  // make room: drop every idle block and try once more
            for (auto it = ctx->cache.begin(); it != ctx->cache.end();)
                if (!it->used) { ctx->cacheBytes -= it->n; (void)hipFree(it->p); it = ctx->cache.erase(it); } else ++it;
            e = hipMalloc(&p, cap);
            if (e != hipSuccess) { p = nullptr; return e; }
        }
        ctx->cache.push_back({p, cap, true});
        ctx->cacheBytes += cap;
        return hipSuccess;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

struct Sub {  // a slice of a DevBuf
    void *p;
    template <class T> T *as() { return static_cast<T *>(p); }
};

// State slots per matrix for children that are kept in full when they are found: the context's setting (1024), cut
// down for very large batches so that the whole state work space stays within 16 GiB.  With fewer slots more
// candidates stay lazy (re-solved when selected): slower, same results.
int eager_states(const kbest_ctx *ctx, int B, int maxRow, int k)
{
    const size_t stride = (size_t)kb::state_stride(maxRow), budget = (size_t)16 << 30;
    const size_t perMatrix = budget / ((size_t)(B > 0 ? B : 1) * stride);
    long long e = (long long)perMatrix - k - ctx->extraStates;
    if (e > ctx->eagerStates) e = ctx->eagerStates;
    if (e < 64) e = ctx->eagerStates < 64 ? ctx->eagerStates : 64;
    return (int)e;
}

}  // namespace

extern "C" {

void kbest_default_opts(kbest_opts *o)
{
    if (!o) return;
    o->maximize = 0;
    o->use_cutoff = 0;
    o->cutoff = 0.0;
    o->flags = 0;
    o->root_col_offset = 0;
    o->root_col_stride = 0;
    o->tie_flags = nullptr;
}

const char *kbest_strerror(int code)
{
    switch (code) {
    case KBEST_OK: return "ok";
    case KBEST_ERR_NO_DEVICE: return "no HIP device available (this engine has no CPU fallback)";
    case KBEST_ERR_BAD_ARG: return "bad argument";
    case KBEST_ERR_UNSUPPORTED: return "problem size not supported by the device kernels";
    case KBEST_ERR_HIP: return "HIP runtime error";
    case KBEST_ERR_NOMEM: return "out of device memory";
    case KBEST_ERR_NOT_RESERVED: return "workspace too small: call kbest_reserve(B, maxRow, k) before the device-pointer entry";
    case KBEST_ERR_INTERNAL: return "internal error in the device kernels (a problem came back with nf < 0)";
    default: return "unknown error";
    }
}

const char *kbest_last_error(const kbest_ctx *ctx)
{
    if (!ctx) return "null context";
    static thread_local std::string copy;  // the caller's view must not change under it when another thread fails
    std::lock_guard<std::mutex> lock(const_cast<kbest_ctx *>(ctx)->errMu);
    copy = ctx->err;
    return copy.c_str();
}

int kbest_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kbest_create(kbest_ctx **out, int device)
{
    if (!out) return KBEST_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return KBEST_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return KBEST_ERR_NO_DEVICE;
    kbest_ctx *ctx = new kbest_ctx;
    ctx->device = device;
    // Stream priorities: the pieces of a large host-entry batch run on four streams of their own, created at the first such
    // call; with the first piece on the highest priority and the later ones below it the pieces FINISH one after the other instead
    // of all at the end, and the host half of a piece (widening its tables) overlaps the later pieces' kernels
    // (KBEST_PIECE_PRIO=0: all equal).  The context's own stream keeps the default priority: a high-priority stream in the
    // process -- even an idle one -- changed how two contexts' launches on the caller's streams overlap (two batches in flight:
    // 1.44 -> 1.79 ms per batch, measured).
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
        const bool prio = !(getenv("KBEST_PIECE_PRIO") && atoi(getenv("KBEST_PIECE_PRIO")) == 0);
        ctx->prioMain = prio ? greatest : 0;
        ctx->prioAux[0] = prio ? (greatest + least) / 2 : 0;
        ctx->prioAux[1] = ctx->prioAux[2] = prio ? least : 0;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->lastEvent, hipEventDisableTiming) != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return KBEST_ERR_NO_DEVICE;
    }
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
        ctx->ldsLimit = lds;
    if (const char *e = getenv("KBEST_NWAVES")) {  // tuning knobs (defaults are the tuned values)
        int w = atoi(e);
        if (w == 4 || w == 8 || w == 12 || w == 16) ctx->nWaves = w;
    }
    if (const char *e = getenv("KBEST_EAGER")) {
        int w = atoi(e);
        if (w >= 0 && w <= 60000) ctx->eagerStates = w;
    }
    if (const char *e = getenv("KBEST_SMALL_NW")) {
        int w = atoi(e);
        if (w == 2 || w == 4 || w == 8 || w == 16) ctx->smallWaves = w;
    }
    ctx->noSmall = getenv("KBEST_NO_SMALL") != nullptr;
    ctx->forceWide = getenv("KBEST_FORCE_WIDE") != nullptr;
    ctx->noLane = getenv("KBEST_NO_LANE") != nullptr;
    ctx->noWideQueue = getenv("KBEST_NO_WIDE_QUEUE") != nullptr;
    ctx->noSplit = getenv("KBEST_NO_SPLIT") != nullptr;
    ctx->noTie = getenv("KBEST_NO_TIE") != nullptr;
    if (const char *e = getenv("KBEST_RELAY")) ctx->relay = atoi(e);
    if (const char *e = getenv("KBEST_RELAY_FIRST")) { const int v = atoi(e); if (v >= 1 && v <= 1023) ctx->relayFirst = v; }
    if (const char *e = getenv("KBEST_RELAY_STEP")) { const int v = atoi(e); if (v >= 1 && v <= 1023) ctx->relayStep = v; }
    if (const char *e = getenv("KBEST_SPLIT")) { const int w = atoi(e); if (w == 2 || w == 4) ctx->splitForce = w; }
    ctx->noTiny = getenv("KBEST_NO_TINY") != nullptr;
    ctx->noBnb = getenv("KBEST_NO_BNB") != nullptr;
    if (const char *e = getenv("KBEST_BNB_SMALL_FROM")) ctx->bnbSmallFrom = atoi(e);
    if (const char *e = getenv("KBEST_ZC_COST")) ctx->zcCost = atoi(e);
    ctx->noReorder = getenv("KBEST_NO_REORDER") != nullptr;
    if (const char *e = getenv("KBEST_PIECES")) { const int w = atoi(e); if (w == 1 || w == 2 || w == 4) ctx->pieces = w; }
    ctx->forceLane = getenv("KBEST_FORCE_LANE") != nullptr;
    if (const char *e = getenv("KBEST_LANE_NW")) { const int w = atoi(e); if (w == 1 || w == 2 || w == 4) ctx->laneNw = w; }
    if (const char *e = getenv("KBEST_LANE_SPEC")) { const int w = atoi(e); if (w >= 1 && w <= kb::LANE_MAX_SPEC) ctx->laneSpec = w; }
    ctx->forceSmall = getenv("KBEST_FORCE_SMALL") != nullptr;
    ctx->noOpt = getenv("KBEST_NO_OPT") != nullptr;
    ctx->noNarrow = getenv("KBEST_NO_NARROW") != nullptr;
    if (const char *e = getenv("KBEST_HOST_THREADS")) ctx->hostThreads = atoi(e);
    if (const char *e = getenv("KBEST_OPT_RHO0")) ctx->optRho0 = (float)atof(e);
    if (const char *e = getenv("KBEST_OPT_RHO1")) ctx->optRho1 = (float)atof(e);
    if (const char *e = getenv("KBEST_OPT_PHI")) ctx->optPhi = (float)atof(e);
    if (const char *e = getenv("KBEST_OPT_KAPPA")) ctx->optKappa = (float)atof(e);
    if (const char *e = getenv("KBEST_OPT_MINPOOL")) ctx->optMinPool = atoi(e);
    ctx->noPoll = getenv("KBEST_NO_POLL") != nullptr;
    if (const char *e = getenv("KBEST_ZC_LIMIT_KB")) ctx->zcLimit = (size_t)atoll(e) << 10;
    ctx->exactRoot = getenv("KBEST_EXACT_ROOT") != nullptr;
    if (const char *e = getenv("KBEST_NO_T0")) ctx->noT0 = atoi(e) != 0;
    if (const char *e = getenv("KBEST_WIDE_NW")) { const int w = atoi(e); if (w == 8 || w == 16) ctx->wideNw = w; }
    if (const char *e = getenv("KBEST_WIDE_TILE")) ctx->wideTile = atoi(e) ? 1 : 0;
    if (const char *e = getenv("KBEST_WIDE_SPEC")) { const int w = atoi(e); if (w >= 1 && w <= kb::WIDE_MAX_SPEC) ctx->wideSpec = w; }
    if (const char *e = getenv("KBEST_SPEC")) {
        int w = atoi(e);
        if (w >= 1 && w <= 16) ctx->spec = w;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->nCU = cus;
    int ldsCU = 0;
    if (hipDeviceGetAttribute(&ldsCU, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess &&
        ldsCU > 0)
        ctx->ldsPerCU = ldsCU;
    *out = ctx;
    return KBEST_OK;
}

int kbest_destroy(kbest_ctx *ctx)
{
    if (!ctx) return KBEST_OK;
    (void)hipSetDevice(ctx->device);
    for (auto &r : ctx->regs) (void)hipHostUnregister(r.host);
    if (ctx->states) (void)hipFree(ctx->states);
    if (ctx->wide) (void)hipFree(ctx->wide);
    if (ctx->wideQueue) (void)hipFree(ctx->wideQueue);
    for (auto &b : ctx->cache) (void)hipFree(b.p);
    if (ctx->pinIn.host) (void)hipHostFree(ctx->pinIn.host);
    if (ctx->pinOut.host) (void)hipHostFree(ctx->pinOut.host);
    if (ctx->pinTab.host) (void)hipHostFree(ctx->pinTab.host);
    if (ctx->stageIn.p) (void)hipFree(ctx->stageIn.p);
    if (ctx->stageOut.p) (void)hipFree(ctx->stageOut.p);
    if (ctx->splitBuf.p) (void)hipFree(ctx->splitBuf.p);
    if (ctx->tieBuf.p) (void)hipFree(ctx->tieBuf.p);
    if (ctx->relayBuf.p) (void)hipFree(ctx->relayBuf.p);
    if (ctx->exactBuf.p) (void)hipFree(ctx->exactBuf.p);
    if (ctx->relayFlags.p) (void)hipFree(ctx->relayFlags.p);
    if (ctx->lastEvent) (void)hipEventDestroy(ctx->lastEvent);
    for (auto &a : ctx->aux)
        if (a) (void)hipStreamDestroy(a);
    if (ctx->hi) (void)hipStreamDestroy(ctx->hi);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return KBEST_OK;
}

int kbest_set_profile_buffer(kbest_ctx *ctx, void *d_buf)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    ctx->prof = static_cast<unsigned long long *>(d_buf);
    return KBEST_OK;
}

// Work space of the general-size kernel: one slot per workgroup of its (persistent) grid.
struct WidePlan {
    int grid;
    size_t cw, states, pool, freeL, perSlot;  // bytes per slot
    int statesPerProblem;
    long long poolStride, freeStride;
};

static WidePlan plan_wide(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k, bool anyCols = false)
{
    auto up = [](size_t x) { return (x + 127) & ~(size_t)127; };
    WidePlan w;
    // pool + children of one round + slack (anyCols: kbest_reserve does not know numCol -- the largest round over numCol <= maxRow)
    const int perRound = anyCols ? (maxRow > 1024 ? maxRow : (kb::WIDE_MAX_SPEC * maxRow < 1024 ? kb::WIDE_MAX_SPEC * maxRow : 1024)) : kb::wide_spec_cap(maxCol, maxRow) * maxCol;
    w.statesPerProblem = k + perRound + 2 + kb::wide_atom_slots(maxRow, anyCols ? maxRow : maxCol);  // + the a-priori threshold's atoms
    w.poolStride = (long long)((k + 1 + 15) & ~15);
    w.freeStride = (long long)((w.statesPerProblem + 31) & ~31);
    w.cw = up((size_t)maxRow * maxRow * 8);
    w.states = (size_t)w.statesPerProblem * (size_t)kb::wide_state_stride(maxRow);
    w.pool = up((size_t)2 * w.poolStride * 8);
    w.freeL = up((size_t)w.freeStride * 4);
    w.perSlot = w.cw + w.states + w.pool + up((size_t)2 * w.poolStride * 4) + w.freeL;
    const size_t budget = (size_t)8 << 30;  // the grid strides over the batch: more slots than this buys nothing
    long long g = (long long)(budget / w.perSlot);
    const long long perCU = maxRow <= 128 ? 3 : (maxRow <= 512 ? 2 : 1);  // resident workgroups per CU (80 VGPRs up to 128 rows, 128 beyond; LDS beyond 512)
    if (g > perCU * ctx->nCU) g = perCU * ctx->nCU;
    if (g > B) g = B;
    if (g < 1) g = 1;
    w.grid = (int)g;
    return w;
}

static int reserve_wide(kbest_ctx *ctx, const WidePlan &w, bool grow)
{
    const size_t need = w.perSlot * (size_t)w.grid + 256;
    if (need <= ctx->wideBytes) return KBEST_OK;
    if (!grow) return fail(ctx, KBEST_ERR_NOT_RESERVED, "general-size work space too small: call kbest_reserve first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->wide) { HIP_TRY(ctx, hipDeviceSynchronize()); (void)hipFree(ctx->wide); ctx->wide = nullptr; ctx->wideBytes = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->wide), need);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_NOMEM, "hipMalloc(general-size work space)", e);
    ctx->wideBytes = need;
    if (!ctx->wideQueue) {
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->wideQueue), 128));
        HIP_TRY(ctx, hipMemset(ctx->wideQueue, 0, 128));
    }
    return KBEST_OK;
}

// Bytes of the LDS kernel's workspace for a (B, maxRow, k) launch, and where the slot table starts in it.
static size_t states_need(const kbest_ctx *ctx, int B, int maxRow, int k, size_t *slotOff)
{
    const size_t nStates = (size_t)B * (size_t)(k + ctx->extraStates + eager_states(ctx, B, maxRow, k)) * (size_t)kb::state_stride(maxRow);
    if (slotOff) *slotOff = (nStates + 127) & ~(size_t)127;
    return nStates + (size_t)B * (size_t)kb::slot_table_stride(k) * 2 + 256;
}

static bool k_fits_fast(const kbest_ctx *ctx, int B, int fastRow, int k, unsigned flags, Shape *shapeOut)
{
    Shape shape = choose_shape(ctx, B, fastRow, k);
    if (flags & KBEST_FLAG_COUNT_PUSHED) shape.spec = 1;  // counting the reference's pushes needs its exact order of splits
    if (shapeOut) *shapeOut = shape;
    return k + ctx->extraStates + ctx->eagerStates <= 65534 && k <= 4 * shape.nWaves * 64 &&
           kb::lds_layout(fastRow, k, shape.spec, shape.nWaves).total <= ctx->ldsLimit;
}

static int ensure_states(kbest_ctx *ctx, size_t need, bool grow);
static int raw_reserve(kbest_ctx *ctx, DevBufRaw &d, size_t need);
static int arena_reserve(kbest_ctx *ctx, kbest_ctx::Arena &a, size_t need);

// per-share result tables [S][B][...] + one shared threshold per matrix of a split launch (split_factor)
struct SplitLayout {
    size_t offGain, offR4C, offNf, offT, bytes;
    SplitLayout(int B, int S, int k, int maxCol)
    {
        auto up = [](size_t x) { return (x + 127) & ~(size_t)127; };
        offGain = 0;
        offR4C = up((size_t)S * B * k * 8);
        offNf = offR4C + up((size_t)S * B * k * maxCol * 4);
        offT = offNf + up((size_t)S * B * 4);
        bytes = offT + up((size_t)B * 8);
    }
};

static int reserve_states(kbest_ctx *ctx, int B, int maxRow, int k, bool grow)
{
    if (maxRow > KBEST_MAX_DIM) maxRow = KBEST_MAX_DIM;
    return ensure_states(ctx, states_need(ctx, B, maxRow, k, nullptr), grow);
}

// Waves per problem of the small-problem kernel (kbest_small.hip): a batch that cannot fill the chip gets a whole
// workgroup of 16 waves (32 half-wave workers) per problem -- latency; a large batch small workgroups -- throughput.
static int small_waves(const kbest_ctx *ctx, int B)
{
    if (ctx->smallWaves > 0) return ctx->smallWaves;
    // (the kernel runs six waves per SIMD: three 8-wave problems per CU at once.  Measured on 200 ... 8 000 KITTI-like frames,
    //  tests/dev/c5_sweep.py: 16 waves win up to one problem per CU, 8 up to ~3 000 frames -- 1 000: 0.54 against 0.60 ms --, 4
    //  beyond; on other rectangular batches, tests/dev/small_shapes.py, ms for 4 / 8 waves: 600 x 32x24, k = 200: 1.09 / 0.86,
    //  800 x 24x12, k = 400: 1.15 / 0.94, but 2 048 x 20x8, k = 50: 0.28 / 0.33, 3 000 x 16x4, k = 20: 0.18 / 0.25 -- short
    //  problems in several generations want the small shape: 8 waves up to two generations of them)
    if (B <= ctx->nCU) return 16;
    if (B <= 6 * ctx->nCU) return 8;
    return 4;
}

static bool small_fits(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k, bool weights, int *nwOut)
{
    if (ctx->noSmall || maxRow > kb::SMALL_MAX_DIM || maxCol > kb::SMALL_MAX_DIM || k > kb::SMALL_MAX_K) return false;
    int nw = small_waves(ctx, B);
    while (nw > 2 && kb::small_lds_layout(maxRow, maxCol, k, nw, weights).total > ctx->ldsLimit)
        nw /= 2;
    if (kb::small_lds_layout(maxRow, maxCol, k, nw, weights).total > ctx->ldsLimit) return false;
    if (nwOut) *nwOut = nw;
    return true;
}

static size_t small_states_need(int B, int maxRow, int maxCol, int k, int nw)
{
    return (size_t)B * (size_t)kb::small_states_per_problem(k, nw, maxCol) * (size_t)kb::small_state_stride(maxRow, maxCol) + 256;
}

// "Launches of up to B problems": a smaller batch takes more waves -- and state slots -- per problem (small_waves), so a
// reservation is the maximum over every tier reachable with B' <= B.
static size_t small_states_need_upto(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k, bool weights)
{
    size_t need = 0;
    const int tiers[3] = {B, B < 6 * ctx->nCU ? B : 6 * ctx->nCU, B < ctx->nCU ? B : ctx->nCU};
    const int shapes[4] = {2, 4, 8, 16};
    for (int t = 0; t < 3; t++)
        for (int i = 0; i < 4; i++) {
            const size_t n = small_states_need(tiers[t], maxRow, maxCol, k, shapes[i]);
            if (kb::small_lds_layout(maxRow, maxCol, k, shapes[i], weights).total <= ctx->ldsLimit && n > need) need = n;
        }
    return need;
}

// Launch shape of the lane-per-child kernel (kbest_lane.hip): waves per problem and hypotheses split per round.  A lane is a
// child, so a round wants about a wave's worth of children per wave: spec * (columns left per hypothesis, ~ half of them).
static Shape lane_shape(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k)
{
    Shape s;
    // measured (kernel ms; tests/dev/lane_sweep.py): a batch of up to four problems per CU is latency -- 4 waves per problem,
    // 8 hypotheses per round (1 024 x 16x16, k = 50: 0.212; 2 waves: 0.235; the 64-row kernel 0.29) --, beyond that throughput --
    // 2 waves (16 384 x 16x16, k = 200: 5.9 ms against 8.8 with 4 waves and 10.2 on the 64-row kernel); 8 hypotheses per round
    // except for short enumerations (k = 10: 4 are 6 % faster)
    s.nWaves = B <= 4 * ctx->nCU ? 4 : 2;
    s.spec = (k <= 16 && B > 4 * ctx->nCU) ? 4 : 8;
    // re-scanned in round 5 (tests/dev/c2_sweep.py, <= 16 rows, ms at 8 / 10 / 12 / 14 / 16 hypotheses per round): 1 024 x 16x16,
    // k = 50: 0.185 / 0.179 / 0.173 / 0.181 / 0.184; 600: 0.174 / 0.158 / 0.163 / .. ; 2 048: 0.232 / 0.224 / 0.227 / ..; 4 096: 0.466 /
    // 0.449 / 0.467 / ..; k = 200: 1 024: 0.525 / 0.480 / 0.454 / 0.432 / 0.414, 8 192: 2.65 / 2.50 / 2.40 / 2.33 / 2.29 -- a long
    // enumeration wants every speculative split it can get (its rounds are what costs), a short one a few more than eight
    if (maxRow <= 16 && s.spec == 8) s.spec = k >= 100 ? 16 : (B <= 4 * ctx->nCU ? 12 : 10);
    if (ctx->laneNw > 0) s.nWaves = ctx->laneNw;
    if (ctx->laneSpec > 0) s.spec = ctx->laneSpec;
    s.lanes = 4;
    while (k > 4 * s.nWaves * 64 && s.nWaves < 4) s.nWaves *= 2;  // the in-place pool merge holds at most 4 entries per thread
    while (s.spec > 1 && kb::lane_lds_layout(maxRow, maxCol, k, s.spec, s.nWaves, s.lanes).total > ctx->ldsLimit) s.spec--;
    (void)B;
    return s;
}

static bool lane_fits(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k, Shape *shapeOut)
{
    if (ctx->noLane || maxRow > kb::LANE_MAX_DIM || maxCol > maxRow) return false;
    const Shape s = lane_shape(ctx, B, maxRow, maxCol, k);
    if (shapeOut) *shapeOut = s;
    return k <= 4 * s.nWaves * 64 && kb::lane_states_per_problem(k, s.spec, maxCol) <= 65534 &&
           kb::lane_lds_layout(maxRow, maxCol, k, s.spec, s.nWaves, s.lanes).total <= ctx->ldsLimit;
}

// bytes of the lane kernel's workspace (saved hypotheses, then the slot -> state table) and where the table starts
static size_t lane_states_need(int B, int maxRow, int maxCol, int k, int spec, size_t *slotOff)
{
    const size_t nStates = (size_t)B * (size_t)kb::lane_states_per_problem(k, spec, maxCol) * (size_t)kb::lane_state_stride(maxRow);
    if (slotOff) *slotOff = (nStates + 127) & ~(size_t)127;
    return ((nStates + 127) & ~(size_t)127) + (size_t)B * (size_t)kb::slot_table_stride(k) * 2 + 256;
}

static int ensure_states(kbest_ctx *ctx, size_t need, bool grow)
{
    if (need <= ctx->statesBytes) return KBEST_OK;
    if (!grow) return fail(ctx, KBEST_ERR_NOT_RESERVED, "hypothesis workspace too small: call kbest_reserve first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->states) { HIP_TRY(ctx, hipDeviceSynchronize()); (void)hipFree(ctx->states); ctx->states = nullptr; ctx->statesBytes = 0; }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->states), need);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_NOMEM, "hipMalloc(state workspace)", e);
    ctx->statesBytes = need;
    return KBEST_OK;
}

static int reserve_for(kbest_ctx *ctx, int B, int maxRow, int k);
static int relay_plan(const kbest_ctx *ctx, int B, int fastRow, int k, unsigned flags, const Shape &shp, size_t *imgOut, double *gensOut = nullptr, bool useCutoff = false);
static int relay_reserve(kbest_ctx *ctx, int B, size_t img, bool grow);

// Work space of the reference-order kernel (kbest_exact.hip): per resident problem ("slot") a padded cost copy and a pool with one
// record per hypothesis the reference would hold -- at most 1 + (k - 1) numCol pushes, plus the one being built.  The grid strides
// over the batch: as many slots as fit 16 GiB, at most eight per CU (one wave each: latency-bound).
struct ExactPlan { int grid, hypPerSlot; size_t slotBytes; };

// (gigabytes: sized to the need, not to the next power of two as raw_reserve does)
static int exact_reserve(kbest_ctx *ctx, size_t need)
{
    DevBufRaw &d = ctx->exactBuf;
    if (need <= d.bytes) return KBEST_OK;
    const size_t cap = (need + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (d.p) { HIP_TRY(ctx, hipDeviceSynchronize()); (void)hipFree(d.p); d.p = nullptr; d.bytes = 0; }
    hipError_t e = hipMalloc(&d.p, cap);
    if (e != hipSuccess) { d.p = nullptr; return fail(ctx, KBEST_ERR_NOMEM, "hipMalloc(reference-order work space)", e); }
    d.bytes = cap;
    return KBEST_OK;
}

static bool plan_exact(const kbest_ctx *ctx, int B, int maxRow, int maxCol, int k, ExactPlan *out)
{
    const long long hyp = 2 + (long long)(k > 1 ? k - 1 : 0) * maxCol;
    if (hyp > 0x7fffffffLL) return false;
    const long long slot = kb::exact_slot_bytes(maxRow, (int)hyp);
    const long long budget = 16LL << 30;
    long long g = budget / (slot > 0 ? slot : 1);
    if (g < 1) return false;
    if (g > 8LL * ctx->nCU) g = 8LL * ctx->nCU;
    if (g > B) g = B;
    out->grid = (int)g;
    out->hypPerSlot = (int)hyp;
    out->slotBytes = (size_t)slot;
    return true;
}

int kbest_reserve_exact(kbest_ctx *ctx, int B, int maxRow, int maxCol, int k)
{
    if (!ctx || B < 0 || maxRow < 1 || maxCol < 1 || maxCol > maxRow || k < 1) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_reserve_exact: bad argument");
    if (maxRow > KBEST_MAX_DIM_EXACT) return fail(ctx, KBEST_ERR_UNSUPPORTED, "numRow > KBEST_MAX_DIM_EXACT");
    if (B == 0) return KBEST_OK;
    ExactPlan pl;
    if (!plan_exact(ctx, B, maxRow, maxCol, k, &pl)) return fail(ctx, KBEST_ERR_NOMEM, "reference-order kernel: one problem's pool of hypotheses does not fit 16 GiB");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return exact_reserve(ctx, pl.slotBytes * (size_t)pl.grid);
}

int kbest_reserve(kbest_ctx *ctx, int B, int maxRow, int k)
{
    if (!ctx || B < 0 || maxRow < 1 || k < 1) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_reserve: bad argument");
    if (maxRow > KBEST_MAX_DIM_EXACT) return fail(ctx, KBEST_ERR_UNSUPPORTED, "numRow > KBEST_MAX_DIM_EXACT");
    if (B == 0) return KBEST_OK;
    if (maxRow > KBEST_MAX_DIM_WIDE) return kbest_reserve_exact(ctx, B, maxRow, maxRow, k);  // (only the reference-order kernel takes such problems)
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    // a launch enumerates k + 1 solutions unless told not to (exact ties, kbest_ties.h): room for either
    int rc = reserve_for(ctx, B, maxRow, k);
    if (rc != KBEST_OK || ctx->noTie) return rc;
    rc = reserve_for(ctx, B, maxRow, k + 1);
    if (rc != KBEST_OK) return rc;
    return raw_reserve(ctx, ctx->tieBuf, (size_t)B * 8);
}

static int reserve_for(kbest_ctx *ctx, int B, int maxRow, int k)
{
    const int fastRow = maxRow < KBEST_MAX_DIM ? maxRow : KBEST_MAX_DIM;
    const bool kFits = k_fits_fast(ctx, B, fastRow, k, 0, nullptr) || k_fits_fast(ctx, B, fastRow, k, KBEST_FLAG_COUNT_PUSHED, nullptr);
    // "launches of up to B problems" (kbest_c.h): a smaller batch may pick another kernel or launch shape than B itself --
    // the small-problem kernel takes more waves (and state slots) per problem as the batch shrinks, and numCol < numRow may
    // change its shape too -- so the reservation is the maximum over every tier reachable with B' <= B.
    if (maxRow <= kb::SMALL_MAX_DIM && !ctx->noSmall && k <= kb::SMALL_MAX_K) {
        const size_t need = small_states_need_upto(ctx, B, maxRow, maxRow, k, false);
        if (need) {
            int rc = ensure_states(ctx, need, true);
            if (rc != KBEST_OK) return rc;
        }
    }
    {   // the lane-per-child kernel: its launch shape (hypotheses split per round -> state slots per problem) follows the batch size
        const int tiers[2] = {B, B < 4 * ctx->nCU ? B : 4 * ctx->nCU};
        for (int t = 0; t < 2; t++) {
            Shape lsh;
            if (lane_fits(ctx, tiers[t], maxRow, maxRow, k, &lsh)) {
                int rc = ensure_states(ctx, lane_states_need(tiers[t], maxRow, maxRow, k, lsh.spec, nullptr), true);
                if (rc != KBEST_OK) return rc;
            }
        }
    }
    if (kFits) {
        // (a small batch of 33 ... 64-row problems may run split: up to four workgroups, i.e. four work spaces, per matrix)
        const int Bs = (ctx->splitForce && !ctx->noSplit && maxRow > 32 && maxRow <= KBEST_MAX_DIM) ? ((4 * B <= ctx->nCU) ? 4 * B : ((2 * B <= ctx->nCU) ? 2 * B : B)) : B;
        int rc = reserve_states(ctx, B, fastRow, k, true);
        if (rc == KBEST_OK) {  // (relay launches: the LDS images of the shape this batch size runs in)
            Shape shp;
            size_t img = 0;
            (void)k_fits_fast(ctx, B, fastRow, k, 0, &shp);
            if (relay_plan(ctx, B, fastRow, k, 0, shp, &img) > 1) rc = relay_reserve(ctx, B, img, true);
        }
        if (rc == KBEST_OK && Bs > B) rc = reserve_states(ctx, Bs, fastRow, k, true);
        if (rc == KBEST_OK && Bs > B) {
            const SplitLayout sl(B, Bs / B, k, maxRow);
            if (sl.bytes > ctx->splitBuf.bytes) rc = raw_reserve(ctx, ctx->splitBuf, sl.bytes);
        }
        if (rc != KBEST_OK) return rc;
    }
    if (!kFits || maxRow > KBEST_MAX_DIM || ctx->forceWide)  // (numCol <= numRow bounds the general-size plan)
        return reserve_wide(ctx, plan_wide(ctx, B, maxRow, maxRow, k, true), true);
    return KBEST_OK;
}

struct DevExtra {  // optional outputs / modes of kbest_assign_batch_f64 (root solution only)
    double *dualU = nullptr, *dualV = nullptr;
    int gainCols = 0;
};

// A launch that is one piece of a larger batch (the host entry sends a big batch through in pieces whose uploads, kernels
// and result traffic overlap): kernel choice, launch shape and workspace are those of the WHOLE batch (`logicalB`), the
// piece uses the workspace slice of its problems (`blockBase` = index of its first problem) and may run concurrently with
// the other pieces on another stream (it is not ordered behind the context's previous launch).
struct SubBatch { int logicalB = 0, blockBase = 0; };
// (A launch with a SubBatch is never a relay: pieces of one batch run side by side on streams of their own, each at most a generation
//  of workgroups; nor is one of a host entry whose kernel writes its tables into HOST memory over the link -- batch_dev_impl's
//  hostTables: a relay's hand-over waits for the write-back of everything its workgroup has stored, and all matrices of a relay finish
//  together at the end instead of one after the other.  1 024 x 64x64, k = 200 through kbest_batch_f64 (tests/dev/host_pieces.sh): four pieces
//  2.5 - 2.66 ms, two 2.65 - 2.70, ONE launch 3.32 - 3.36, one launch as a relay 3.1 - 3.6.)

// Order this launch (on stream s) behind the previous launch of the context when that ran on another stream: both
// use the context's one hypothesis workspace.  Called with ctx->mu held.  A caller-owned stream may be destroyed by its
// owner at any time after our launch, so it is never touched again: the event that marks the end of a launch on such a
// stream is recorded right behind the launch (Launched, below), while the stream is certainly alive; only the context's
// own stream -- which lives as long as the context -- gets its event recorded lazily, here, when a switch happens.
static bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return s && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

static int order_behind_last(kbest_ctx *ctx, hipStream_t s)
{
    // Under stream capture (kbest_c.h allows the asynchronous entries inside a graph capture) nothing of the cross-stream
    // bookkeeping may happen: an event recorded on a capturing stream becomes a node of the graph, and another stream that
    // later waits on it would be pulled into the capture (or fail).  A captured launch is therefore NOT ordered against the
    // context's other streams -- a context whose launches are captured is used on that one stream (kbest_c.h).
    if (capturing(s)) return KBEST_OK;
    if (ctx->haveLast && ctx->lastStream != s) {
        if (ctx->lastStream == ctx->stream) HIP_TRY(ctx, hipEventRecord(ctx->lastEvent, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->lastEvent, 0));
    }
    ctx->lastStream = s;
    ctx->haveLast = true;
    return KBEST_OK;
}

// Scope guard of an asynchronous entry: whatever was enqueued on a caller-owned stream `s` is marked by the context's
// event when the entry returns (see order_behind_last).
struct Launched {
    kbest_ctx *ctx;
    hipStream_t s;
    ~Launched() { if (s != ctx->stream && !capturing(s)) (void)hipEventRecord(ctx->lastEvent, s); }
};

// One matrix over several workgroups (64-row kernel).  When a batch leaves CUs idle -- at most half as many 33 ... 64-row
// square problems as CUs -- every matrix is alone on a CU and bound by the latency of its own rounds (128 x 64x64, k = 200:
// 1.04 ms where 1 024 take 2.7).  S workgroups can take the root's subtrees in turn (columns c % S == share), share their
// thresholds through HBM and have a k-way merge assemble the k best (kbest_merge.hip).  Built, bit-exact -- and SLOWER in
// every case measured (tests/dev/split_check.py: 128 x 64x64 1.05 -> 1.27 ms with two workgroups per matrix, one matrix alone
// 0.68 -> 0.83 / 0.96 with two / four; 48x48, k = 100 the same picture): a share needs as many rounds as the whole matrix (its
// front advances through the same gains), its a-priori thresholds come from half the atoms, and the shared threshold lags by
// a round.  Off unless KBEST_SPLIT = 2 / 4 asks for it (the parity test and the measurements use that).
static int split_factor(const kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, int k, bool uniform, bool plain)
{
    if (!ctx->splitForce || ctx->noSplit || !uniform || !plain || maxRow != maxCol || maxRow <= 32 || maxRow > KBEST_MAX_DIM || k < 50 ||
        opts->root_col_stride > 1 || (opts->flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_PRUNE | KBEST_FLAG_RECT_ROOT |
                                                     KBEST_FLAG_NO_SHIFT | KBEST_FLAG_EXACT_ROOT | KBEST_FLAG_TABLES_I8)))
        return 1;
    return ctx->splitForce * B <= ctx->nCU ? ctx->splitForce : ((2 * B <= ctx->nCU) ? 2 : 1);
}

// grow: the host-pointer entries (which synchronise anyway) let the workspace grow on demand; the asynchronous
// device-pointer entry never allocates or synchronises -- it needs kbest_reserve up front.
// Relay launches of the 64-row kernel (kbest_engine.hip): pieces per matrix for a batch of B matrices in launch shape `shp`, and
// the bytes of one LDS image.  A launch of a few generations of resident workgroups ends with the slot whose matrices add up to
// the most (25 % of a C4 launch's slot-time is idle, NOTES 10.3); pieces a fraction of a lifetime long let the slots even out.
static int relay_plan(const kbest_ctx *ctx, int B, int fastRow, int k, unsigned flags, const Shape &shp, size_t *imgOut, double *gensOut, bool useCutoff)
{
    if (gensOut) *gensOut = 0.0;
    const int ldsB = kb::lds_layout(fastRow, k, shp.spec, shp.nWaves).total;
    if (imgOut) *imgOut = ((size_t)ldsB + 16 + 127) & ~(size_t)127;
    if ((flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_PRUNE | KBEST_FLAG_RECT_ROOT | KBEST_FLAG_NO_SHIFT)) || k < 16) return 1;
    // (the relay instantiations of the kernel: 4 / 8 / 12 waves -- launch_nw, kbest_engine.hip)
    if (!(shp.nWaves == 4 || shp.nWaves == 8 || shp.nWaves == 12)) return 1;
    int perCU = ctx->ldsPerCU / (ldsB > 0 ? ldsB : 1);
    const int byWaves = (6 * 4) / shp.nWaves;  // six waves per SIMD (80 VGPRs)
    perCU = perCU < byWaves ? perCU : byWaves;
    const double gens = (double)B / (double)((perCU > 0 ? perCU : 1) * ctx->nCU);
    if (ctx->relay < 0 && gens <= 1.0) return 1;  // (KBEST_RELAY forces the pieces on any batch: tests)
    // measured (tests/dev/relay_sweep.py, ms plain / best relay; NOTES 10.6): 64x64, k = 200: 600 matrices (1.2 generations) 1.367 /
    // 1.176, 700: 1.411 / 1.204, 1 024 (2.0): 1.820 / 1.570, 1 536: 2.591 / 2.298, 3 072 (6.0): 4.746 / 4.522; 32x32: 2 048: 2.161 /
    // 1.884, 3 072: 3.011 / 2.776, 4 096: 3.581 / 3.238, 8 192: 6.584 / 6.331.  Three pieces are at or within 1 % of the best of
    // 2 ... 8 everywhere up to six generations (a hand-over costs ~11 us of a slot: the image out and in, a workgroup's start, a cold
    // L1); beyond that two.
    if (gensOut) *gensOut = gens;
    // A batch of BARELY more than one generation is where a plain launch is worst -- 513 matrices on 512 slots take two lifetimes --
    // and where fine slices pay most: up to 1.8 generations four pieces at quarters of k (tests/dev/relay_edge.sh, ms plain / 3 pieces at
    // 3/8, 3/4 / 4 pieces at quarters: 513 x 64x64 1.195 / 1.027 / 0.939, 530: 1.273 / 1.082 / 1.022, 600: 1.406 / 1.200 / 1.141, 700: 1.433 /
    // 1.211 / 1.164, 768: 1.547 / 1.287 / 1.242; 1 100 x 32x32 (8 waves): 1.350 / 1.161 / 1.101).
    int P = ctx->relay >= 0 ? ctx->relay : (gens <= 1.8 ? 4 : (gens <= 6.5 ? 3 : (gens <= 10.0 ? 2 : 1)));
    // (with a cutoff a matrix may end long before its k-th solution: its later pieces' workgroups then start only to find it
    //  finished -- ~5 us of a slot each.  tests/dev/relay_cutoff.py, plain / three pieces: 6 000 x 32x32 with 1.2 solutions per matrix
    //  inside the cutoff 0.276 / 0.305 ms, with 6: 0.723 / 0.747; 2 048 x 64x64 with 57: 1.684 / 1.636, with 197: 3.27 / 2.98 -- two
    //  pieces risk half of that)
    if (ctx->relay < 0 && useCutoff && P > 2) P = 2;
    return P > 8 ? 8 : (P < 1 ? 1 : P);
}

static int relay_reserve(kbest_ctx *ctx, int B, size_t img, bool grow)
{
    const size_t need = (size_t)B * img, needF = (size_t)B * 16;  // (three arrays of words in quarters of the buffer)
    if (need <= ctx->relayBuf.bytes && needF <= ctx->relayFlags.bytes) return KBEST_OK;
    if (!grow) return KBEST_ERR_NOT_RESERVED;
    // a captured relay launch holds the addresses of these buffers: growing (= freeing) them under it is refused
    if (ctx->relayCaptured && ctx->relayBuf.p)
        return fail(ctx, KBEST_ERR_BAD_ARG, "the relay work space cannot grow while a captured launch of this context holds it: reserve for the largest batch before capturing");
    int rc = raw_reserve(ctx, ctx->relayBuf, need);
    if (rc != KBEST_OK) return rc;
    if (needF > ctx->relayFlags.bytes) {
        rc = raw_reserve(ctx, ctx->relayFlags, needF);
        if (rc != KBEST_OK) return rc;
        HIP_TRY(ctx, hipMemset(ctx->relayFlags.p, 0, ctx->relayFlags.bytes));
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    return KBEST_OK;
}

static bool tie_mode(const kbest_ctx *ctx, const kbest_opts *opts, bool rootOnly)
{
    return !ctx->noTie && !rootOnly && opts->root_col_stride <= 1 &&
           !(opts->flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_PRUNE | KBEST_FLAG_RECT_ROOT | KBEST_FLAG_NO_SHIFT | KBEST_FLAG_NO_TIE_CHECK |
                            KBEST_FLAG_REFERENCE_ORDER));
}

static int batch_dev_impl(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                          const int32_t *d_nRow, const int32_t *d_nCol, const double *d_cost,
                          const int64_t *d_costOff, int k, int32_t *d_row4col, int32_t *d_col4row,
                          double *d_gain, int32_t *d_nf, int64_t *d_pushed, void *stream, bool grow,
                          const DevExtra *extra = nullptr, const SubBatch *sub = nullptr, bool hostTables = false)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (!opts || B < 0 || k < 1 || maxCol < 1 || maxRow < maxCol || !d_cost || !d_row4col || !d_gain || !d_nf)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64_dev: bad argument");
    if ((d_nRow == nullptr) != (d_nCol == nullptr))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64_dev: give both nRow and nCol or neither");
    if (maxRow > KBEST_MAX_DIM_EXACT) return fail(ctx, KBEST_ERR_UNSUPPORTED, "numRow > KBEST_MAX_DIM_EXACT");
    const bool tabI8 = (opts->flags & KBEST_FLAG_TABLES_I8) != 0;
    if (tabI8 && extra) return fail(ctx, KBEST_ERR_BAD_ARG, "KBEST_FLAG_TABLES_I8: k-best entries only");
    if (tabI8 && maxRow > 127) return fail(ctx, KBEST_ERR_UNSUPPORTED, "KBEST_FLAG_TABLES_I8: numRow > 127 does not fit int8 tables");
    if (B == 0) return KBEST_OK;
    // The reference's own order of operations (kbest_exact.hip): asked for (KBEST_FLAG_REFERENCE_ORDER), or the only kernel for the size.
    if (!extra && ((opts->flags & KBEST_FLAG_REFERENCE_ORDER) || maxRow > KBEST_MAX_DIM_WIDE)) {
        if (opts->root_col_stride > 1) return fail(ctx, KBEST_ERR_BAD_ARG, "root-subtree sharding: not with the reference-order kernel");
        ExactPlan pl;
        const int LBx = sub ? sub->logicalB : B;
        if (!plan_exact(ctx, LBx, maxRow, maxCol, k, &pl)) return fail(ctx, KBEST_ERR_NOMEM, "reference-order kernel: one problem's pool of hypotheses does not fit 16 GiB");
        if (sub) return fail(ctx, KBEST_ERR_INTERNAL, "a piece of a batch on the reference-order kernel");
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
        int rc = order_behind_last(ctx, s);
        if (rc != KBEST_OK) return rc;
        const Launched mark{ctx, s};
        const size_t need = pl.slotBytes * (size_t)pl.grid;
        if (need > ctx->exactBuf.bytes) {
            if (!grow) return fail(ctx, KBEST_ERR_NOT_RESERVED, "reference-order work space too small: call kbest_reserve_exact first");
            rc = exact_reserve(ctx, need);
            if (rc != KBEST_OK) return rc;
        }
        kb::ExactParams ep;
        memset(&ep, 0, sizeof(ep));
        ep.cost = d_cost;
        ep.costOff = reinterpret_cast<const long long *>(d_costOff);
        ep.nRow = d_nRow;
        ep.nCol = d_nCol;
        ep.B = B;
        ep.maxRow = maxRow;
        ep.maxCol = maxCol;
        ep.ldRow = maxRow;
        ep.ldCol = maxCol;
        ep.k = k;
        ep.maximize = opts->maximize;
        ep.useCutoff = opts->use_cutoff;
        ep.flags = opts->flags;
        ep.cutoff = opts->cutoff;
        ep.row4col = d_row4col;
        ep.col4row = d_col4row;
        ep.gain = d_gain;
        ep.nf = d_nf;
        ep.pushed = (opts->flags & KBEST_FLAG_COUNT_PUSHED) ? reinterpret_cast<long long *>(d_pushed) : nullptr;
        ep.work = static_cast<unsigned char *>(ctx->exactBuf.p);
        ep.hypPerSlot = pl.hypPerSlot;
        const hipError_t e = kb::launch_kbest_exact(ep, pl.grid, s);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "reference-order kbest kernel launch", e);
        ctx->lastRoute = KBEST_ROUTE_EXACT;
        if (opts->tie_flags) HIP_TRY(ctx, hipMemsetAsync(opts->tie_flags, 0, (size_t)B * 4, s));  // (this mode IS the reference's order: nothing to flag)
        return KBEST_OK;
    }
    // Two kernels share the work.  The LDS kernel (kbest_engine.hip) takes every problem of up to KBEST_MAX_DIM
    // rows as long as the candidate pool for k fits its LDS; the general-size kernel (kbest_wide.hip) takes the
    // rest: larger problems of a mixed batch (shapes on the device: both are launched, each skips the other's
    // problems), a uniform batch of larger problems, or any shape when k is beyond the LDS pool.
    const int fastRow = maxRow < KBEST_MAX_DIM ? maxRow : KBEST_MAX_DIM;
    const int fastCol = maxCol < fastRow ? maxCol : fastRow;
    const int LB = sub ? sub->logicalB : B;          // the batch that decides kernel, shape and workspace
    const size_t base = sub ? (size_t)sub->blockBase : 0;  // first problem of this piece in that batch
    // Exact ties (kbest_ties.h): the kernels enumerate ONE solution more than the tables hold -- kT slots per problem, kernel-side
    // k = kT + 1 -- and order runs of equal gains canonically.  Not in the modes that reproduce the reference's own order of
    // operations (push counting, the unpruned run, the root-only entries) nor under root-subtree sharding (the merge orders ties).
    const bool tieMode = tie_mode(ctx, opts, extra != nullptr);
    const int kT = k;
    // Where k sits exactly at a limit of the kernel that takes it (k = 4 x waves x 64 of the 64-row kernel's pool merge,
    // SMALL_MAX_K, an LDS pool that is just full) k + 1 no longer fits that kernel.  The SYNCHRONOUS entries (grow) then take the
    // kernel that does take k + 1 -- slower at exactly that k, but a tie at slot k is seen and completed: the one answer --; the
    // ASYNCHRONOUS entry keeps its kernel and its speed, runs WITHOUT the extra solution and flags its problems
    // KBEST_TIE_UNCHECKED (a tie at slot k would not be seen; runs inside the tables are ordered as ever).
    bool extraSol = tieMode;
    if (tieMode && !grow) {
        const bool fastA = k_fits_fast(ctx, LB, fastRow, kT, opts->flags, nullptr), fastB = k_fits_fast(ctx, LB, fastRow, kT + 1, opts->flags, nullptr);
        const bool laneA = lane_fits(ctx, LB, maxRow, maxCol, kT, nullptr), laneB = lane_fits(ctx, LB, maxRow, maxCol, kT + 1, nullptr);
        const bool smallA = small_fits(ctx, LB, maxRow, maxCol, kT, false, nullptr), smallB = small_fits(ctx, LB, maxRow, maxCol, kT + 1, false, nullptr);
        if ((fastA && !fastB) || (laneA && !laneB) || (smallA && !smallB)) extraSol = false;
    }
    if (extraSol) k = k + 1;
    int32_t *d_tie = (tieMode && opts->tie_flags) ? opts->tie_flags + base : nullptr;
    Shape shape;
    const bool kFits = k_fits_fast(ctx, LB, fastRow, k, opts->flags, &shape);
    const bool forceWide = ctx->forceWide && !extra;  // test hook: everything through the general-size kernel
    const bool runFast = !forceWide && kFits && (maxRow <= KBEST_MAX_DIM || d_nRow != nullptr);
    const bool runWide = forceWide || !kFits || maxRow > KBEST_MAX_DIM;
    if (extra && runWide) return fail(ctx, KBEST_ERR_UNSUPPORTED, "assign2D / shortestPathCPP entry: numRow > KBEST_MAX_DIM");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    if (!sub) {
        int rc = order_behind_last(ctx, s);
        if (rc != KBEST_OK) return rc;
    }
    const Launched mark{ctx, sub ? ctx->stream : s};
    double *d_tieGain = nullptr;
    if (extraSol) {
        if ((size_t)LB * 8 > ctx->tieBuf.bytes) {
            if (!grow) return fail(ctx, KBEST_ERR_NOT_RESERVED, "tie work space too small: call kbest_reserve first");
            const int rc = raw_reserve(ctx, ctx->tieBuf, (size_t)LB * 8);
            if (rc != KBEST_OK) return rc;
        }
        d_tieGain = static_cast<double *>(ctx->tieBuf.p) + base;
    }
    // the launch behind the enumeration: runs of equal gains into the canonical order, the tie flags (kbest_ties.h)
    auto finish = [&]() -> int {
        if (!tieMode) return KBEST_OK;
        hipError_t e = kb::launch_finish_tables(d_nf, d_nRow, d_nCol, B, kT, maxCol, maxRow, d_row4col, d_col4row, d_gain, tabI8, d_tieGain, d_tie,
                                                false, s, extraSol ? 0 : KBEST_TIE_UNCHECKED, !hostTables);
        return e == hipSuccess ? KBEST_OK : fail(ctx, KBEST_ERR_HIP, "tie-order kernel launch", e);
    };
    // Problems of up to 32 rows: the small-problem kernel (half-wave workers, implicit zero columns).  The modes that
    // need the reference's exact order of splits (push counting), no pruning, subtree sharding or the duals of the
    // padded formulation stay on the 64-row kernel.
    // Which of the two wins was measured: the small kernel on rectangular problems (implicit zero columns:
    // a third of the Dijkstra steps) and on batches that cannot fill the chip (latency: 32 workers per problem, two
    // barriers per round); the 64-row kernel with its hand-written step loop on large batches of square problems.
    // Dense batches of <= 32-row problems that fill the chip: the lane-per-child kernel (kbest_lane.hip).  Plain enumeration
    // only (no push counting, no unpruned mode, no duals): those stay on the 64-row kernel.
    Shape lsh;
    // Measured against the 64-row kernel (tests/dev/lane_sweep.py): up to 16 rows 1.2 - 1.7 x faster from B = 600 to 16 384 and
    // k = 10 to 200; 17 - 32 rows (twice the rows per lane) 10 % faster only where the batch is small enough for 4 waves per
    // problem and the enumeration long, else up to 20 % slower (4 096 x 32x32, k = 200: 5.4 ms against 4.9).  Batches that
    // cannot fill the chip (B <= 2 CUs) stay on the small-problem kernel (16 waves per problem: 0.33 against 0.40 ms at
    // 256 x 16x16, k = 200), rectangular ones too (implicit zero columns).
    // Since the kernels enumerate in a column order of their own (DESIGN.md section 2 point 8; the small-problem kernel does
    // not: nothing to gain with its free rows), uniform SQUARE batches no longer go to the small-problem kernel at any size:
    // 16x16, k = 50: 0.124 - 0.172 ms on the lane kernel against 0.125 - 0.228 from 1 to 512 problems; 32x32, k = 200:
    // 0.36 - 0.67 ms on the 64-row kernel against 0.50 - 1.02.
    const bool squareU = maxCol == maxRow && d_nRow == nullptr;
    const bool laneWins = ctx->forceLane || (maxCol == maxRow && !ctx->forceSmall &&
                                             ((maxRow <= 16 && (squareU || LB > 2 * ctx->nCU)) ||
                                              (LB > 2 * ctx->nCU && LB <= 4 * ctx->nCU && k >= 100)));
    if (!extra && !forceWide && laneWins && !(opts->flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_PRUNE | KBEST_FLAG_RECT_ROOT |
                                                                 KBEST_FLAG_NO_SHIFT | KBEST_FLAG_EXACT_ROOT)) &&
        lane_fits(ctx, LB, maxRow, maxCol, k, &lsh)) {
        size_t slotOff = 0;
        int rc = ensure_states(ctx, lane_states_need(LB, maxRow, maxCol, k, lsh.spec, &slotOff), grow);
        if (rc != KBEST_OK) return rc;
        kb::Params p;
        memset(&p, 0, sizeof(p));
        p.cost = d_cost;
        p.costOff = reinterpret_cast<const long long *>(d_costOff);
        p.nRow = d_nRow;
        p.nCol = d_nCol;
        p.maxRow = maxRow;
        p.maxCol = maxCol;
        p.ldRow = maxRow;
        p.ldCol = maxCol;
        p.k = k;
        p.maximize = opts->maximize;
        p.useCutoff = opts->use_cutoff;
        p.flags = opts->flags | (ctx->noReorder ? KBEST_FLAG_NO_REORDER : 0u);
        p.cutoff = opts->cutoff;
        p.rootColOffset = opts->root_col_offset;
        p.rootColStride = opts->root_col_stride;
        p.row4col = d_row4col;
        p.col4row = d_col4row;
        p.gain = d_gain;
        p.nf = d_nf;
        p.pushed = reinterpret_cast<long long *>(d_pushed);
        p.stateStride = kb::lane_state_stride(maxRow);
        p.statesPerProblem = kb::lane_states_per_problem(k, lsh.spec, maxCol);
        p.states = ctx->states + base * (size_t)p.statesPerProblem * (size_t)p.stateStride;
        p.lazyStates = p.statesPerProblem;
        p.spec = lsh.spec;
        p.prof = ctx->prof;
        p.slotSid = reinterpret_cast<unsigned short *>(ctx->states + slotOff) + base * (size_t)kb::slot_table_stride(k);
        p.kTab = kT;
        p.tieGain = d_tieGain;
        hipError_t e = kb::launch_kbest_lane(p, B, lsh.nWaves, lsh.lanes, s);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "lane-per-child kbest kernel launch", e);
        ctx->lastRoute = KBEST_ROUTE_LANE | (extraSol ? KBEST_ROUTE_EXTRA : 0);
        return finish();
    }
    int snw = 0;
    const bool smallWins = ctx->forceSmall || maxCol < maxRow || (LB <= 2 * ctx->nCU && !squareU);
    if (!extra && !forceWide && smallWins && !(opts->flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_PRUNE)) &&
        opts->root_col_stride <= 1 && small_fits(ctx, LB, maxRow, maxCol, k, false, &snw)) {
        int rc = ensure_states(ctx, small_states_need(LB, maxRow, maxCol, k, snw), grow);
        if (rc != KBEST_OK) return rc;
        kb::SmallParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.cost = d_cost;
        sp.costOff = reinterpret_cast<const long long *>(d_costOff);
        sp.nRow = d_nRow;
        sp.nCol = d_nCol;
        sp.maxRow = maxRow;
        sp.maxCol = maxCol;
        sp.ldRow = maxRow;
        sp.ldCol = maxCol;
        sp.k = k;
        sp.maximize = opts->maximize;
        sp.useCutoff = opts->use_cutoff;
        sp.cutoff = opts->cutoff;
        sp.row4col = d_row4col;
        sp.col4row = d_col4row;
        sp.tabI8 = tabI8 ? 1 : 0;
        sp.gain = d_gain;
        sp.nf = d_nf;
        sp.stateStride = kb::small_state_stride(maxRow, maxCol);
        sp.statesPerProblem = kb::small_states_per_problem(k, snw, maxCol);
        sp.states = ctx->states + base * (size_t)sp.statesPerProblem * (size_t)sp.stateStride;
        sp.prof = ctx->prof;
        sp.kTab = kT;
        sp.tieGain = d_tieGain;
        hipError_t e = kb::launch_kbest_small(sp, B, snw, s);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "small-problem kbest kernel launch", e);
        ctx->lastRoute = KBEST_ROUTE_SMALL | (extraSol ? KBEST_ROUTE_EXTRA : 0);
        return finish();
    }

    if (sub && runWide) return fail(ctx, KBEST_ERR_INTERNAL, "a piece of a batch on the general-size kernel");
    ctx->lastRoute = (runFast ? KBEST_ROUTE_FAST : 0) | (runWide ? KBEST_ROUTE_WIDE : 0) | (extraSol ? KBEST_ROUTE_EXTRA : 0);
    if (runFast) {
        // (a split launch merges per-share lists of k: it runs without the extra solution; the merge orders ties by the assignment)
        const int S = (sub || tieMode) ? 1 : split_factor(ctx, opts, B, maxRow, maxCol, k, d_nRow == nullptr && d_costOff == nullptr, extra == nullptr);
        const int PB = LB * S;  // workgroups = problems of the launch as the kernel sees them
        Shape shp = shape;
        if (S > 1 && !k_fits_fast(ctx, PB, fastRow, k, opts->flags, &shp)) return fail(ctx, KBEST_ERR_INTERNAL, "split launch shape");
        int rc = reserve_states(ctx, PB, fastRow, k, grow);
        if (rc != KBEST_OK) return rc;
        size_t slotOff = 0;
        (void)states_need(ctx, PB, fastRow, k, &slotOff);
        const SplitLayout sl(B, S, k, maxCol);
        if (S > 1) {
            if (sl.bytes > ctx->splitBuf.bytes) {
                if (!grow) return fail(ctx, KBEST_ERR_NOT_RESERVED, "split work space too small: call kbest_reserve first");
                rc = raw_reserve(ctx, ctx->splitBuf, sl.bytes);
                if (rc != KBEST_OK) return rc;
            }
            HIP_TRY(ctx, hipMemsetAsync(static_cast<char *>(ctx->splitBuf.p) + sl.offT, 0xFF, (size_t)B * 8, s));  // no threshold yet
        }
        char *sb = static_cast<char *>(ctx->splitBuf.p);
        kb::Params p;
        memset(&p, 0, sizeof(p));
        p.cost = d_cost;
        p.costOff = reinterpret_cast<const long long *>(d_costOff);
        p.nRow = d_nRow;
        p.nCol = d_nCol;
        p.maxRow = fastRow;
        p.maxCol = fastCol;
        p.ldRow = maxRow;
        p.ldCol = maxCol;
        p.k = k;
        p.maximize = opts->maximize;
        p.useCutoff = opts->use_cutoff;
        p.flags = opts->flags | (ctx->exactRoot ? KBEST_FLAG_EXACT_ROOT : 0u) | (ctx->noT0 ? KBEST_FLAG_NO_T0 : 0u) |
                  (ctx->noReorder ? KBEST_FLAG_NO_REORDER : 0u);
        p.cutoff = opts->cutoff;
        p.rootColOffset = opts->root_col_offset;
        p.rootColStride = opts->root_col_stride;
        p.row4col = S > 1 ? reinterpret_cast<int32_t *>(sb + sl.offR4C) : d_row4col;
        p.col4row = S > 1 ? nullptr : d_col4row;
        p.gain = S > 1 ? reinterpret_cast<double *>(sb + sl.offGain) : d_gain;
        p.nf = S > 1 ? reinterpret_cast<int32_t *>(sb + sl.offNf) : d_nf;
        p.pushed = reinterpret_cast<long long *>(d_pushed);
        p.stateStride = kb::state_stride(fastRow);
        p.statesPerProblem = k + ctx->extraStates + eager_states(ctx, PB, fastRow, k);
        p.states = ctx->states + base * (size_t)p.statesPerProblem * (size_t)p.stateStride;
        p.lazyStates = k + ctx->extraStates;
        p.spec = shp.spec;
        p.prof = ctx->prof;
        p.slotSid = reinterpret_cast<unsigned short *>(ctx->states + slotOff) + base * (size_t)kb::slot_table_stride(k);
        p.dualU = extra ? extra->dualU : nullptr;
        p.dualV = extra ? extra->dualV : nullptr;
        p.gainCols = extra ? extra->gainCols : 0;
        p.kTab = kT;
        p.tieGain = d_tieGain;
        p.split = S;
        p.splitB = B;
        p.sharedT = S > 1 ? reinterpret_cast<unsigned long long *>(sb + sl.offT) : nullptr;
        // optimistic bounds: the quantile of the pool a node is split against (host model: tests/dev/proto_tickets.cpp; kernel:
        // struct Opt).  Measured (kernel ms, off -> on, one box, interleaved): 4 096 x 32x32, k = 200 (4 waves, 4 hypotheses per
        // round, no a-priori thresholds there) 3.87 -> 3.67 at 0.85 (0.8: 3.81, 0.9: 3.76, 0.75: 4.12); 1 024 x 64x64 (12 x 12, where
        // the a-priori thresholds already bound the early rounds) 1.865 -> 1.856 at 0.8, 1.90 at 0.7, 2.16 at 0.55: nothing to
        // gain there, and the extra rounds of re-splits cost -- off in the shapes that run the a-priori thresholds.
        {
            const bool t0Shape = shp.nWaves >= 8;  // (kbest_engine.hip: t0On needs 8 waves; OPT_SHAPE compiles the mechanism out there)
            // Where it pays was measured on dense square batches (tests/dev/opt_ab.py, 4 waves x 4, off -> on): 4 096 x 32x32, k = 200
            // -5 %, k = 400 -1.6 %, 28x28, k = 100 -2.4 %; but 24x24, k = 200 +2.2 %, 17x17, k = 200 +12 % (the early pool of a small
            // problem is a poor sample of where its k-th best will lie: 27 re-splits per matrix in the host model against 4 at 32 rows),
            // 32x32, k = 50 +1.5 %, k = 20 +5.9 % (short enumerations: nothing to save, the bookkeeping remains).  Hence on from 28
            // rows and k = 100 only; KBEST_OPT_RHO0 forces it anywhere.
            const bool optShape = !t0Shape && fastRow >= 28 && k >= 100;
            const float rho0 = ctx->optRho0 >= 0.0f ? ctx->optRho0 : (optShape ? 0.85f : 2.0f);
            if (ctx->noOpt || rho0 >= 1.0f) p.optRho0 = 2.0f;
            else {
                p.optRho0 = rho0;
                p.optRho1 = ctx->optRho1 >= 0.0f ? ctx->optRho1 : rho0;
                const float phi = ctx->optPhi > 0.0f ? ctx->optPhi : 1.0f;
                p.optSlope = (p.optRho1 - p.optRho0) / (phi * (float)k);
                p.optKappa = (double)ctx->optKappa;
                p.optMinPool = ctx->optMinPool > 1 ? ctx->optMinPool : 2;
            }
        }
        // Relay: a launch of a few generations ends with the slot whose matrices add up to the most; enumerating every matrix in
        // pieces (workgroups that hand the matrix' LDS on through HBM) lets the slots even out (kbest_engine.hip; NOTES 10.3,
        // 10.6).  Whole launches of the plain enumeration only, where the batch is 1.2 ... 6 generations of resident workgroups.
        size_t relayImg = 0;
        double relayGens = 0.0;
        const int relayP = (!sub && !extra && S == 1 && !hostTables && opts->root_col_stride <= 1) ? relay_plan(ctx, B, fastRow, k, opts->flags, shp, &relayImg, &relayGens, opts->use_cutoff != 0) : 1;
        if (relayP > 1) {
            rc = relay_reserve(ctx, B, relayImg, grow);
            if (rc == KBEST_OK) {
                p.relayP = relayP;
                p.relayB = B;
                // (where the pieces end, in solutions -- tests/dev/relay_sweep.py, three pieces: the 12-wave shape's matrices -- 64 rows --
                //  up to two generations: cuts at 3/8 and 3/4 of k: 1.563 ms against 1.580 at 3/8, 11/16 and 1.602 at 1/2, 3/4; everything
                //  else at 5/8 and 7/8: 4 096 x 32x32 3.16 ms against 3.21 at 5/8, 13/16 and 3.27 at 1/2, 3/4.  The last pieces are the
                //  launch's last generation: short ones let it drain evenly, and the late solutions are the cheap ones)
                const bool wide12 = shp.nWaves == 12 && relayGens <= 2.2, quarters = relayGens <= 1.8 && ctx->relay < 0;
                p.relayFirst = ctx->relayFirst > 0 ? ctx->relayFirst : (quarters ? (relayP >= 4 ? 256 : 512) : (wide12 ? 384 : 640));
                p.relayStep = ctx->relayStep > 0 ? ctx->relayStep
                              : (ctx->relay >= 0 ? (1024 - p.relayFirst) / (relayP - 1)   // (a forced count: even steps, every piece hands over)
                                                 : (quarters ? 256 : (wide12 ? 384 : 256)));
                if (ctx->relayDirty.exchange(false)) {  // (an earlier entry of this context failed: the words may not be zero)
                    const hipError_t ez = kb::launch_zero_words(static_cast<unsigned *>(ctx->relayFlags.p), (long long)(ctx->relayFlags.bytes / 4), s);
                    if (ez != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "relay words: zeroing kernel launch", ez);
                }
                if (capturing(s)) ctx->relayCaptured = true;
                ctx->relayLaunches++;
                ctx->lastRoute |= KBEST_ROUTE_RELAY;
                p.relayBuf = static_cast<unsigned char *>(ctx->relayBuf.p);
                p.relayStride = (long long)relayImg;
                p.relayFlag = static_cast<unsigned *>(ctx->relayFlags.p);
                p.relayClaim = p.relayFlag + ctx->relayFlags.bytes / 16;  // (quarters of the buffer, wherever a smaller batch ends)
                p.relayGone = p.relayFlag + ctx->relayFlags.bytes / 8;
            } else if (rc != KBEST_ERR_NOT_RESERVED) {
                return rc;
            }  // (not reserved: an asynchronous entry never allocates -- the launch runs without the relay)
        }
        hipError_t e = kb::launch_kbest(p, B * S, shp.nWaves, s);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "kbest kernel launch", e);
        if (S > 1) {  // the global k best of every matrix from its shares' lists
            kb::MergeParams mp;
            memset(&mp, 0, sizeof(mp));
            mp.gain = reinterpret_cast<const unsigned char *>(sb + sl.offGain);
            mp.row4col = reinterpret_cast<const unsigned char *>(sb + sl.offR4C);
            mp.nf = reinterpret_cast<const unsigned char *>(sb + sl.offNf);
            mp.shardStride = (long long)B * k * 8;
            mp.strideR4C = (long long)B * k * maxCol * 4;
            mp.strideNf = (long long)B * 4;
            mp.nShard = S;
            mp.k = k;
            mp.maxCol = maxCol;
            mp.ldCol = maxCol;
            mp.maximize = opts->maximize;
            mp.outGain = d_gain;
            mp.outRow4col = d_row4col;
            mp.outNf = d_nf;
            mp.outCol4row = d_col4row;
            mp.ldRow = maxRow;
            e = kb::launch_merge_topk(mp, B, s);
            if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "merge kernel launch", e);
        }
    }
    if (runWide) {
        const WidePlan w = plan_wide(ctx, B, maxRow, maxCol, k);
        const int nwMin = maxRow > 512 ? 4 : 8;  // (beyond 512 rows: 16 rows per lane, four waves per problem -- the waves' LDS working sets)
        if (kb::wide_lds_layout(maxRow, maxCol, false, nwMin, 1).total > ctx->ldsLimit)
            return fail(ctx, KBEST_ERR_UNSUPPORTED, "problem too large for the general-size kernel's LDS");
        // hypotheses split per round: counting the reference's pushes needs the reference's exact order of splits (1)
        auto spec_for = [&](int nwv) {
            int sp = (opts->flags & KBEST_FLAG_COUNT_PUSHED) ? 1 : (ctx->wideSpec > 0 ? ctx->wideSpec : kb::wide_spec(maxCol, nwv, k));
            const int cap = kb::wide_spec_cap(maxCol, maxRow);
            return sp > cap ? cap : sp;
        };
        // up to ~128 rows the square cost copy fits LDS next to the waves' working sets: a cost column then comes
        // from LDS instead of L2 at every Dijkstra step.  It costs residency (one workgroup per CU), so only a batch
        // that leaves CUs idle anyway takes it; such a batch also gets 16 waves per problem instead of 8.
        int nw = (B <= ctx->nCU) ? 16 : 8;
        if (ctx->wideNw) nw = ctx->wideNw;
        if (maxRow > 512) nw = 4;
        int spec = spec_for(nw);
        bool tile = maxRow <= 128 && (ctx->wideTile >= 0 ? ctx->wideTile == 1 : B <= ctx->nCU) &&
                    kb::wide_lds_layout(maxRow, maxCol, true, nw, spec).total <= ctx->ldsLimit;
        if (!tile && nw > 8 && kb::wide_lds_layout(maxRow, maxCol, false, nw, spec).total > ctx->ldsLimit) { nw = 8; spec = spec_for(nw); }
        while (spec > 1 && kb::wide_lds_layout(maxRow, maxCol, tile, nw, spec).total > ctx->ldsLimit) spec--;
        int rc = reserve_wide(ctx, w, grow);
        if (rc != KBEST_OK) return rc;
        kb::WideParams p;
        p.cost = d_cost;
        p.costOff = reinterpret_cast<const long long *>(d_costOff);
        p.nRow = d_nRow;
        p.nCol = d_nCol;
        p.B = B;
        p.maxRow = maxRow;
        p.maxCol = maxCol;
        p.ldRow = maxRow;
        p.ldCol = maxCol;
        p.minRows = runFast ? KBEST_MAX_DIM + 1 : 0;
        p.tile = tile ? 1 : 0;
        p.nw = nw;
        p.spec = spec;
        p.k = k;
        p.maximize = opts->maximize;
        p.useCutoff = opts->use_cutoff;
        p.flags = opts->flags | (ctx->noT0 ? KBEST_FLAG_NO_T0 : 0u) | (ctx->noReorder ? KBEST_FLAG_NO_REORDER : 0u);
        p.cutoff = opts->cutoff;
        p.rootColOffset = opts->root_col_offset;
        p.rootColStride = opts->root_col_stride;
        p.row4col = d_row4col;
        p.col4row = d_col4row;
        p.gain = d_gain;
        p.nf = d_nf;
        p.pushed = reinterpret_cast<long long *>(d_pushed);
        unsigned char *base = ctx->wide;
        const size_t G = (size_t)w.grid;
        p.Cw = reinterpret_cast<double *>(base);                  base += w.cw * G;
        p.cwStride = (long long)(w.cw / 8);
        p.states = base;                                          base += w.states * G;
        p.stateStride = kb::wide_state_stride(maxRow);
        p.statesPerProblem = w.statesPerProblem;
        p.poolG = reinterpret_cast<double *>(base);               base += w.pool * G;
        p.poolStride = w.poolStride;
        p.poolS = reinterpret_cast<int *>(base);                  base += ((size_t)2 * w.poolStride * 4 + 127) / 128 * 128 * G;
        p.freeList = reinterpret_cast<int *>(base);
        p.freeStride = w.freeStride;
        p.prof = ctx->prof;
        p.kTab = kT;
        p.tieGain = d_tieGain;
        p.queue = (ctx->noWideQueue || capturing(s)) ? nullptr : ctx->wideQueue;  // (a captured launch: fixed stride -- nothing to go wrong on a replay)
        hipError_t e = kb::launch_kbest_wide(p, w.grid, s);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "general-size kbest kernel launch", e);
    }
    return finish();
}

int kbest_batch_f64_dev(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                        const int32_t *d_nRow, const int32_t *d_nCol, const double *d_cost,
                        const int64_t *d_costOff, int k, int32_t *d_row4col, int32_t *d_col4row,
                        double *d_gain, int32_t *d_nf, int64_t *d_pushed, void *stream)
{
    if (opts && (opts->flags & (KBEST_FLAG_RECT_ROOT | KBEST_FLAG_NO_SHIFT)))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64_dev: internal flag");
    return batch_dev_impl(ctx, opts, B, maxRow, maxCol, d_nRow, d_nCol, d_cost, d_costOff, k, d_row4col, d_col4row,
                          d_gain, d_nf, d_pushed, stream, false);
}

// device address of [p, p + n) when the whole range lies inside a registered host buffer, else nullptr
static void *mapped(kbest_ctx *ctx, const void *p, size_t n)
{
    std::lock_guard<std::mutex> lock(ctx->regMu);
    const char *q = static_cast<const char *>(p);
    for (const auto &r : ctx->regs)
        if (q >= r.host && q + n <= r.host + r.bytes) return r.dev + (q - r.host);
    return nullptr;
}

int kbest_register_host_buffer(kbest_ctx *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || bytes == 0) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_register_host_buffer: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "hipHostRegister", e);
    void *dev = nullptr;
    e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) { (void)hipHostUnregister(ptr); return fail(ctx, KBEST_ERR_HIP, "hipHostGetDevicePointer", e); }
    std::lock_guard<std::mutex> lock(ctx->regMu);
    ctx->regs.push_back({static_cast<char *>(ptr), static_cast<char *>(dev), bytes});
    return KBEST_OK;
}

int kbest_unregister_host_buffer(kbest_ctx *ctx, void *ptr)
{
    if (!ctx || !ptr) return KBEST_ERR_BAD_ARG;
    {
        std::lock_guard<std::mutex> lock(ctx->regMu);
        auto it = ctx->regs.begin();
        for (; it != ctx->regs.end(); ++it)
            if (it->host == ptr) break;
        if (it == ctx->regs.end()) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_unregister_host_buffer: not registered");
        ctx->regs.erase(it);
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);  // nothing may still be writing there
    HIP_TRY(ctx, hipHostUnregister(ptr));
    return KBEST_OK;
}

}  // extern "C"

static inline size_t outBytesHint(int B, int k, int maxRow, int maxCol) { return (size_t)B * k * ((size_t)maxRow + maxCol) * 4; }

// Runs of equal gains of ONE problem's tables in host memory into the canonical order (kbest_ties.h: row4col lexicographic in the
// reference's column order) -- what the finishing launch does for tables in device memory; for tables the kernels wrote into HOST
// memory (the narrow-staged and the registered-buffer paths of the host entry) the launch only reports KBEST_TIE_INSIDE and the
// entry orders them here.  T: int32 or int8 tables; col4row may be null; M / N: the problem's columns / rows; ld*: the tables' widths.
template <class T> static void order_ties_host(const double *gain, T *row4col, T *col4row, int nf, int M, int N, int ldCol, int ldRow)
{
    std::vector<int> idx;
    std::vector<T> tmpR, tmpC;
    for (int s = 0; s + 1 < nf;) {
        int e = s + 1;
        while (e < nf && gain[e] == gain[s]) e++;
        const int L = e - s;
        if (L > 1) {
            idx.resize(L);
            for (int i = 0; i < L; i++) idx[i] = s + i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) {
                const T *ra = row4col + (size_t)a * ldCol, *rb = row4col + (size_t)b * ldCol;
                for (int c = 0; c < M; c++)
                    if (ra[c] != rb[c]) return ra[c] < rb[c];
                return false;
            });
            tmpR.assign(row4col + (size_t)s * ldCol, row4col + (size_t)e * ldCol);
            if (col4row) tmpC.assign(col4row + (size_t)s * ldRow, col4row + (size_t)e * ldRow);
            for (int i = 0; i < L; i++) {
                const int from = idx[i] - s;
                if (from == i) continue;
                memcpy(row4col + (size_t)(s + i) * ldCol, tmpR.data() + (size_t)from * ldCol, (size_t)M * sizeof(T));
                if (col4row) memcpy(col4row + (size_t)(s + i) * ldRow, tmpC.data() + (size_t)from * ldRow, (size_t)N * sizeof(T));
            }
        }
        s = e;
    }
}

// ... for every problem of a host-entry call that the launch flagged KBEST_TIE_INSIDE (flags: the device buffer the launch wrote)
static int order_flagged_host(kbest_ctx *ctx, const int32_t *d_flags, int B, int k, int maxRow, int maxCol, const int32_t *nRow, const int32_t *nCol,
                              const double *gain, void *row4col, void *col4row, const int32_t *nf, bool i8)
{
    std::vector<int32_t> fl((size_t)B);
    HIP_TRY(ctx, hipMemcpy(fl.data(), d_flags, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++) {
        if (!(fl[b] & KBEST_TIE_INSIDE)) continue;
        const int n = nf[b] < 0 ? 0 : (nf[b] > k ? k : nf[b]);
        const int M = nCol ? nCol[b] : maxCol, N = nRow ? nRow[b] : maxRow;
        if (i8)
            order_ties_host<signed char>(gain + (size_t)b * k, static_cast<signed char *>(row4col) + (size_t)b * k * maxCol,
                                         col4row ? static_cast<signed char *>(col4row) + (size_t)b * k * maxRow : nullptr, n, M, N, maxCol, maxRow);
        else
            order_ties_host<int32_t>(gain + (size_t)b * k, static_cast<int32_t *>(row4col) + (size_t)b * k * maxCol,
                                     col4row ? static_cast<int32_t *>(col4row) + (size_t)b * k * maxRow : nullptr, n, M, N, maxCol, maxRow);
    }
    return KBEST_OK;
}

// The host-buffer entry.  `keep` (kbest_multi.cpp): the result tables are staged in the CALLER's device buffers -- a device's
// packed slice of the multi-device global table -- and stay there after they have been copied back, so that the all-gather
// can follow; everything else (pieces, uploads, copies back) is the single-device path.
int kbest_batch_f64_keep(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                         const int32_t *nRow, const int32_t *nCol, const double *cost,
                         const int64_t *costOff, int k, int32_t *row4col, int32_t *col4row, double *gain,
                         int32_t *nf, int64_t *pushed, const kb::KeepTables *keep)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (!opts || B < 0 || k < 1 || maxCol < 1 || maxRow < maxCol || !cost || !row4col || !gain || !nf)  // (col4row NULL: not wanted)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64: bad argument");
    if (opts->flags & (KBEST_FLAG_RECT_ROOT | KBEST_FLAG_NO_SHIFT)) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64: internal flag");
    if ((nRow == nullptr) != (nCol == nullptr))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64: give both nRow and nCol or neither");
    if (B == 0) return KBEST_OK;
    size_t nCost = 0;
    if (nRow) {
        for (int b = 0; b < B; b++) {
            if (nCol[b] < 1 || nRow[b] < nCol[b] || nRow[b] > maxRow || nCol[b] > maxCol)
                return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64: shape out of range (need 1 <= numCol <= numRow <= maxRow)");
            const size_t end = (costOff ? (size_t)costOff[b] : (size_t)b * maxRow * maxCol) + (size_t)nRow[b] * nCol[b];
            if (end > nCost) nCost = end;
        }
    } else {
        if (costOff) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64: costOff needs per-problem shapes");
        nCost = (size_t)B * maxRow * maxCol;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int kk = tie_mode(ctx, opts, false) ? k + 1 : k;  // what the kernels enumerate (exact ties, kbest_ties.h)
    const size_t nR4C = (size_t)B * k * maxCol, nC4R = (size_t)B * k * maxRow, nG = (size_t)B * k;
    const bool tabI8 = (opts->flags & KBEST_FLAG_TABLES_I8) != 0;  // row4col / col4row are int8 tables behind the int32_t pointers
    const size_t esz = tabI8 ? 1 : 4;
    if (tabI8 && maxRow > 127) return fail(ctx, KBEST_ERR_UNSUPPORTED, "KBEST_FLAG_TABLES_I8: numRow > 127 does not fit int8 tables");
    // Result tables in registered (pinned, device-mapped) host memory (kbest_register_host_buffer) are written THERE by the
    // kernels, spread over the whole run (the 64-row kernel emits a slot's tables as soon as the slot is final), so that
    // nothing is left to copy when the last matrix ends; otherwise they are staged in device buffers and copied back.
    char *mR4C = static_cast<char *>(mapped(ctx, row4col, nR4C * esz));
    char *mC4R = col4row ? static_cast<char *>(mapped(ctx, col4row, nC4R * esz)) : nullptr;
    double *mGain = static_cast<double *>(mapped(ctx, gain, nG * 8));
    int32_t *mNf = static_cast<int32_t *>(mapped(ctx, nf, (size_t)B * 4));
    const bool direct = mR4C && (mC4R || !col4row) && mGain && mNf && !pushed && !keep;
    double *mCost = static_cast<double *>(mapped(ctx, cost, nCost * 8));
    const bool pinnedCost = mCost != nullptr;
    // Registered cost blocks with registered result tables: the LDS kernels read every cost block exactly once, into their
    // tile, so they take it from the caller's memory over the link themselves -- no upload in front of the first workgroup,
    // later generations' reads hide behind the running ones (1 024 x 64x64, k = 200: 3.74 -> 3.33 ms per call; 3.00 with
    // int8 tables).  Not for the general-size kernel (it re-reads costs from memory at every step), and not together with
    // copies back of pageable tables (measured slower: they share the link).
    // (the reference-order kernel -- KBEST_FLAG_REFERENCE_ORDER -- runs whole batches from device memory: no pieces, no narrow staging)
    const bool refOrder = (opts->flags & KBEST_FLAG_REFERENCE_ORDER) != 0;
    const bool zcCost = pinnedCost && direct && ctx->zcCost && maxRow <= KBEST_MAX_DIM && !ctx->forceWide && !refOrder &&
                        k_fits_fast(ctx, B, maxRow < KBEST_MAX_DIM ? maxRow : KBEST_MAX_DIM, kk, opts->flags, nullptr);
    // Narrow staging: the reference's int32 tables are 107 MB for 1 024 x 64x64, k = 200, and the link, not the kernel, set the time
    // of this entry (3.1 ms against 1.8).  Every index fits a byte and col4row is the inverse of row4col on a square problem, so
    // the kernels write row4col as BYTES into pinned staging memory (13 MB, as the slots become final), in pieces, and host
    // threads widen a piece into the caller's row4col / col4row while the GPU works on the next one.  Uniform square batches of
    // up to 64 rows (every row has a column: the inverse is complete); everything else takes the path below.
    const bool narrow = (!keep || keep->row4col8) && !tabI8 && !pushed && !nRow && !costOff && maxRow == maxCol && maxRow <= KBEST_MAX_DIM && !ctx->forceWide && !refOrder &&
                        !ctx->noNarrow && outBytesHint(B, k, maxRow, maxCol) >= ((size_t)8 << 20) &&
                        k_fits_fast(ctx, B, maxRow, kk, opts->flags, nullptr);
    if (narrow) {
        std::lock_guard<std::mutex> narrowLock(ctx->narrowMu);
        const size_t offG = (nR4C + 63) & ~(size_t)63, offN = offG + nG * 8, tabBytes = offN + (size_t)B * 4;
        {
            std::lock_guard<std::recursive_mutex> lock(ctx->mu);
            int rc0 = arena_reserve(ctx, ctx->pinTab, tabBytes);
            if (rc0 != KBEST_OK) return rc0;
        }
        int nth = ctx->hostThreads > 0 ? ctx->hostThreads - 1 : (int)std::thread::hardware_concurrency() - 1;
        nth = nth > 15 ? 15 : (nth < 0 ? 0 : nth);
        HostPool *pool = host_pool(nth);
        signed char *h8 = static_cast<signed char *>(ctx->pinTab.host);
        char *d8 = static_cast<char *>(ctx->pinTab.dev);
        const double *hG = reinterpret_cast<const double *>(h8 + offG);
        const int32_t *hN = reinterpret_cast<const int32_t *>(h8 + offN);
        DevBuf dCostN;
        // Registered (pinned) cost blocks: uploaded by the copy engine PIECE AFTER PIECE on a stream of their own, each piece's kernel
        // behind its own upload -- the first piece's blocks are up after a quarter of the transfer and the rest arrives under the
        // running kernels.  (The kernels reading the blocks in place over the link -- KBEST_ZC_COST=2, what round 4 did here -- keeps
        // the whole first generation of workgroups waiting for 16 MB of tiles: 2.95 ms per call of 1 024 x 64x64 against 2.67 with
        // plain pageable buffers; all four pieces' copies at once share the link and the first piece is up no sooner: 3.16.)
        const bool zc = pinnedCost && ctx->zcCost == 2;
        const bool chained = pinnedCost && !zc && ctx->zcCost != 0;
        if (!zc) HIP_TRY(ctx, dCostN.alloc(ctx, nCost * 8));
        if (chained && !ctx->copy) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
        hipEvent_t up[4] = {nullptr, nullptr, nullptr, nullptr};
        const double *devC = zc ? mCost : dCostN.as<double>();
        const int nP = (B >= 4 * ctx->nCU) ? (ctx->pieces > 0 ? ctx->pieces : 4) : 1;
        for (int i = 0; i < 3 && nP > 1; i++)
            if (!ctx->aux[i]) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->aux[i], hipStreamNonBlocking, ctx->prioAux[i]));
        if (nP > 1 && !ctx->hi) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->hi, hipStreamNonBlocking, ctx->prioMain));
        kbest_opts o8 = *opts;
        o8.flags |= KBEST_FLAG_TABLES_I8;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}, start = nullptr;
        hipStream_t st[4] = {nP > 1 ? ctx->hi : ctx->stream, ctx->aux[0], ctx->aux[1], ctx->aux[2]};
        int rc = KBEST_OK;
        std::unique_lock<std::recursive_mutex> pieceLock(ctx->mu);  // the context is held from the first piece to the last
        rc = order_behind_last(ctx, ctx->stream);
        if (rc == KBEST_OK && nP > 1) {
            if (hipEventCreateWithFlags(&start, hipEventDisableTiming) != hipSuccess || hipEventRecord(start, ctx->stream) != hipSuccess)
                rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: event", hipGetLastError());
            for (int c = 0; c < nP && rc == KBEST_OK; c++)
                if (hipStreamWaitEvent(st[c], start, 0) != hipSuccess) rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: wait", hipGetLastError());
        }
        const size_t per = (size_t)maxRow * maxCol;
        // the host half of a piece: row4col widened, col4row = its inverse, gains and counts copied -- by the context's host threads
        auto widen_piece = [&](int c) {
            const int b0 = (int)((long long)B * c / nP), nb = (int)((long long)B * (c + 1) / nP) - b0;
            if (hipEventSynchronize(ev[c]) != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: synchronize", hipGetLastError()); return; }
            const int chunk = 2, nTask = (nb + chunk - 1) / chunk;
            const std::function<void(int)> widen = [&](int t) {
                const int lo = b0 + t * chunk, hi = lo + chunk < b0 + nb ? lo + chunk : b0 + nb;
                for (int b = lo; b < hi; b++) {
                    const signed char *src = h8 + (size_t)b * k * maxCol;
                    int32_t *r4 = row4col + (size_t)b * k * maxCol;
                    int32_t *c4 = col4row ? col4row + (size_t)b * k * maxRow : nullptr;
                    for (int sl = 0; sl < k; sl++) {
                        const signed char *s8 = src + (size_t)sl * maxCol;
                        int32_t *r = r4 + (size_t)sl * maxCol;
                        for (int j = 0; j < maxCol; j++) r[j] = (int32_t)s8[j];
                        if (c4) {
                            int32_t *cc = c4 + (size_t)sl * maxRow;
                            for (int j = 0; j < maxRow; j++) cc[j] = -1;
                            for (int j = 0; j < maxCol; j++)
                                if (s8[j] >= 0) cc[s8[j]] = j;
                        }
                    }
                    memcpy(gain + (size_t)b * k, hG + (size_t)b * k, (size_t)k * 8);
                    nf[b] = hN[b];
                }
            };
            pool->run(nTask, widen);
        };
        for (int c = 0; c < nP && rc == KBEST_OK; c++) {
            const int b0 = (int)((long long)B * c / nP), nb = (int)((long long)B * (c + 1) / nP) - b0;
            hipError_t e = hipSuccess;
            if (keep && keep->stamps && c == 0) keep->stamps[0] = kb::now_s();
            if (!zc) {
                if (chained) {
                    if (c == 0 && start) e = hipStreamWaitEvent(ctx->copy, start, 0);  // (behind the context's previous launch, as the pieces are)
                    if (e == hipSuccess) e = hipMemcpyAsync(dCostN.as<double>() + (size_t)b0 * per, cost + (size_t)b0 * per, (size_t)nb * per * 8, hipMemcpyHostToDevice, ctx->copy);
                    if (e == hipSuccess) e = hipEventCreateWithFlags(&up[c], hipEventDisableTiming);
                    if (e == hipSuccess) e = hipEventRecord(up[c], ctx->copy);
                    if (e == hipSuccess) e = hipStreamWaitEvent(st[c], up[c], 0);
                } else if (pinnedCost) e = hipMemcpyAsync(dCostN.as<double>() + (size_t)b0 * per, cost + (size_t)b0 * per, (size_t)nb * per * 8, hipMemcpyHostToDevice, st[c]);
                else e = hipMemcpy(dCostN.as<double>() + (size_t)b0 * per, cost + (size_t)b0 * per, (size_t)nb * per * 8, hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: upload", e); break; }
            const SubBatch sub{B, b0};
            // the piece's tables: pinned staging (the kernel writes over the link), or -- keep -- the caller's device buffers
            int32_t *pR = keep ? reinterpret_cast<int32_t *>(keep->row4col8 + (size_t)b0 * k * maxCol) : reinterpret_cast<int32_t *>(d8 + (size_t)b0 * k * maxCol);
            double *pG = keep ? keep->gain + (size_t)b0 * k : reinterpret_cast<double *>(d8 + offG) + (size_t)b0 * k;
            int32_t *pN = keep ? keep->nf + b0 : reinterpret_cast<int32_t *>(d8 + offN) + b0;
            rc = batch_dev_impl(ctx, &o8, nb, maxRow, maxCol, nullptr, nullptr, devC + (size_t)b0 * per, nullptr, k, pR, nullptr, pG, pN, nullptr,
                                st[c], true, nullptr, nP > 1 ? &sub : nullptr, !keep);  // (!keep: tables in host memory -- never a relay, see SubBatch)
            if (rc != KBEST_OK) break;
            e = kb::launch_fill_unused(pN, nullptr, nullptr, nb, k, maxCol, maxRow, pR, nullptr, pG, true, st[c]);
            if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "fill kernel launch", e); break; }
            if (keep) {
                // the device keeps row4col as its slice of the global table holds it (int32: widened here; bytes: as written); the bytes,
                // the gains and the counts go home
                const size_t nEnt = (size_t)nb * k * maxCol;
                if (!keep->keepI8) e = kb::launch_widen_i8(reinterpret_cast<const signed char *>(pR), keep->row4col + (size_t)b0 * k * maxCol, (long long)nEnt, st[c]);
                // (by copy kernels on the piece's stream into the host-mapped staging: kbest_merge.hip, launch_copy_words)
                if (e == hipSuccess && nEnt % 4 == 0) e = kb::launch_copy_words(pR, d8 + (size_t)b0 * k * maxCol, (long long)nEnt, st[c]);
                else if (e == hipSuccess) e = hipMemcpyAsync(h8 + (size_t)b0 * k * maxCol, pR, nEnt, hipMemcpyDeviceToHost, st[c]);
                if (e == hipSuccess) e = kb::launch_copy_words(pG, d8 + offG + (size_t)b0 * k * 8, (long long)nb * k * 8, st[c]);
                if (e == hipSuccess) e = kb::launch_copy_words(pN, d8 + offN + (size_t)b0 * 4, (long long)nb * 4, st[c]);
                if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: copy back", e); break; }
            }
            if (keep && keep->stamps && c == 0) keep->stamps[1] = kb::now_s();
            if (hipEventCreateWithFlags(&ev[c], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[c], st[c]) != hipSuccess) {
                rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: event", hipGetLastError());
                break;
            }
            if (st[c] != ctx->stream && hipStreamWaitEvent(ctx->stream, ev[c], 0) != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: wait", hipGetLastError()); break; }
        }
        pieceLock.unlock();
        // (all pieces are in flight first: doing the host half of piece c - 1 before piece c + 1 is uploaded and launched was measured
        //  slower, 2.47 -> 2.97 ms -- the host then waits for a piece that shares the GPU with its successor)
        for (int c = 0; c < nP && rc == KBEST_OK; c++) widen_piece(c);
        for (int c = 0; c < nP; c++) {
            const hipError_t e = hipStreamSynchronize(st[c]);
            if (e != hipSuccess && rc == KBEST_OK) rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: synchronize", e);
        }
        if (chained) (void)hipStreamSynchronize(ctx->copy);
        for (int c = 0; c < 4; c++) {
            if (ev[c]) (void)hipEventDestroy(ev[c]);
            if (up[c]) (void)hipEventDestroy(up[c]);
        }
        if (start) (void)hipEventDestroy(start);
        if (rc != KBEST_OK) return rc;
        // (the byte tables went into pinned HOST staging: runs of equal gains were reported, not ordered -- here, in the caller's tables)
        if (!keep && opts->tie_flags && tie_mode(ctx, opts, false)) {
            rc = order_flagged_host(ctx, opts->tie_flags, B, k, maxRow, maxCol, nullptr, nullptr, gain, row4col, col4row, nf, false);
            if (rc != KBEST_OK) return rc;
        }
        for (int b = 0; b < B; b++)
            if (nf[b] < 0) return fail(ctx, nf[b] == -1 ? KBEST_ERR_UNSUPPORTED : KBEST_ERR_INTERNAL, "kbest_batch_f64: a problem came back with nf < 0");
        return KBEST_OK;
    }
    DevBuf dCost, dOff, dNR, dNC, dR4C, dC4R, dGain, dNf, dPushed;
    if (!zcCost) HIP_TRY(ctx, dCost.alloc(ctx, nCost * 8));
    const double *devCost = zcCost ? mCost : dCost.as<double>();
    if (keep && (tabI8 || !keep->row4col || (col4row && !keep->col4row) || (keep->keepI8 && !keep->row4col8)))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_batch_f64_keep: bad device tables");
    if (!direct && !keep) {
        HIP_TRY(ctx, dR4C.alloc(ctx, nR4C * esz));
        if (col4row) HIP_TRY(ctx, dC4R.alloc(ctx, nC4R * esz));
        HIP_TRY(ctx, dGain.alloc(ctx, nG * 8));
        HIP_TRY(ctx, dNf.alloc(ctx, (size_t)B * 4));
    }
    if (!direct && pushed) HIP_TRY(ctx, dPushed.alloc(ctx, (size_t)B * 8));
    // the device-side staging of the tables: the caller's (keep) or the context's recycled blocks
    char *sR4C = keep ? reinterpret_cast<char *>(keep->row4col) : dR4C.as<char>();
    char *sC4R = keep ? reinterpret_cast<char *>(keep->col4row) : dC4R.as<char>();
    double *sGain = keep ? keep->gain : dGain.as<double>();
    int32_t *sNf = keep ? keep->nf : dNf.as<int32_t>();
    // (tables addressed in bytes: esz per entry)
    char *oR4C = direct ? mR4C : sR4C, *oC4R = !col4row ? nullptr : (direct ? mC4R : sC4R);
    int32_t *oNf = direct ? mNf : sNf;
    double *oGain = direct ? mGain : sGain;
    // A batch that is large in problems and in output bytes goes through the GPU in PIECES on separate streams: the launch
    // shape and the workspace are those of the whole batch (SubBatch), so the pieces' workgroups fill the chip exactly as one
    // launch of the whole batch would -- but a piece starts as soon as ITS cost blocks are up, and its tables cross PCIe
    // while later pieces still run.  1 024 x 64x64, k = 200 (33 MB in, 107 MB out): pageable buffers 5.7 ms in one piece,
    // 4.4 in two half-size launches (round 2), now four pieces; registered buffers 3.7 ms in one piece.
    const size_t outBytes = (nR4C + (col4row ? nC4R : 0)) * esz + nG * 8;
    const int fastRow = maxRow < KBEST_MAX_DIM ? maxRow : KBEST_MAX_DIM;
    const bool canPiece = !costOff && maxRow <= KBEST_MAX_DIM && !ctx->forceWide && !refOrder && B >= 4 * ctx->nCU && outBytes >= ((size_t)32 << 20) &&
                          k_fits_fast(ctx, B, fastRow, kk, opts->flags, nullptr);
    const int nPiece = canPiece ? (ctx->pieces > 0 ? ctx->pieces : ((zcCost && direct) ? 1 : 4)) : 1;
    if (nPiece > 1)
        for (int i = 0; i < 3; i++)
            if (!ctx->aux[i]) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->aux[i], hipStreamNonBlocking, ctx->prioAux[i]));
    if (nRow) {
        HIP_TRY(ctx, dNR.alloc(ctx, (size_t)B * 4));
        HIP_TRY(ctx, dNC.alloc(ctx, (size_t)B * 4));
        HIP_TRY(ctx, hipMemcpy(dNR.p, nRow, (size_t)B * 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dNC.p, nCol, (size_t)B * 4, hipMemcpyHostToDevice));
    }
    if (costOff) {
        HIP_TRY(ctx, dOff.alloc(ctx, (size_t)B * 8));
        HIP_TRY(ctx, hipMemcpy(dOff.p, costOff, (size_t)B * 8, hipMemcpyHostToDevice));
    }
    int rc = KBEST_OK;
    hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr}, start = nullptr;
    if (nPiece > 1 && !ctx->hi) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->hi, hipStreamNonBlocking, ctx->prioMain));
    hipStream_t st[4] = {nPiece > 1 ? ctx->hi : ctx->stream, ctx->aux[0], ctx->aux[1], ctx->aux[2]};
    // A pieced launch holds the context from its first piece to its last: the pieces run on the context's auxiliary streams on
    // slices of the one hypothesis workspace, and another thread's launch on this context (kbest_batch_f64_dev on a stream of
    // its own) must neither slip in between them nor overlap them -- it is ordered behind ctx->stream, which waits for every
    // piece below.
    std::unique_lock<std::recursive_mutex> pieceLock(ctx->mu, std::defer_lock);
    if (nPiece > 1) {
        // the pieces run on other streams than the context's previous (possibly still running) launch: order them behind it
        pieceLock.lock();
        rc = order_behind_last(ctx, ctx->stream);
        if (rc == KBEST_OK && (hipEventCreateWithFlags(&start, hipEventDisableTiming) != hipSuccess || hipEventRecord(start, ctx->stream) != hipSuccess))
            rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: event", hipGetLastError());
        for (int c = 0; c < nPiece && rc == KBEST_OK; c++)
            if (hipStreamWaitEvent(st[c], start, 0) != hipSuccess) rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: wait", hipGetLastError());
    }
    const size_t per = (size_t)maxRow * maxCol;
    for (int c = 0; c < nPiece && rc == KBEST_OK; c++) {
        const int b0 = (int)((long long)B * c / nPiece), nb = (int)((long long)B * (c + 1) / nPiece) - b0;
        const size_t cOff = costOff ? 0 : (size_t)b0 * per, cLen = (nPiece == 1) ? nCost : (size_t)nb * per;
        hipError_t e;
        // (registered cost blocks: asynchronous copies, all pieces' at once -- they share the link, so the first piece's kernel
        //  starts ~0.3 ms into the call; chaining the copies with events, or blocking copies piece by piece, measured slower:
        //  3.9 - 4.1 ms per call against 3.7)
        if (keep && keep->stamps && c == 0) keep->stamps[0] = kb::now_s();
        if (zcCost) e = hipSuccess;
        else if (pinnedCost) e = hipMemcpyAsync(dCost.as<double>() + cOff, cost + cOff, cLen * 8, hipMemcpyHostToDevice, st[c]);
        else e = hipMemcpy(dCost.as<double>() + cOff, cost + cOff, cLen * 8, hipMemcpyHostToDevice);  // (complete on return; earlier pieces run meanwhile)
        if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: upload", e); break; }
        const SubBatch sub{B, b0};
        rc = batch_dev_impl(ctx, opts, nb, maxRow, maxCol, nRow ? dNR.as<int32_t>() + b0 : nullptr, nRow ? dNC.as<int32_t>() + b0 : nullptr,
                            costOff ? devCost : devCost + (size_t)b0 * per, costOff ? dOff.as<int64_t>() + b0 : nullptr,
                            k, reinterpret_cast<int32_t *>(oR4C + (size_t)b0 * k * maxCol * esz),
                            oC4R ? reinterpret_cast<int32_t *>(oC4R + (size_t)b0 * k * maxRow * esz) : nullptr, oGain + (size_t)b0 * k, oNf + b0,
                            pushed ? dPushed.as<int64_t>() + b0 : nullptr, st[c], true, nullptr, nPiece > 1 ? &sub : nullptr, direct);
        if (rc != KBEST_OK) break;
        if (keep && keep->stamps && c == 0) keep->stamps[1] = kb::now_s();
        // slots beyond nf are never written by the kernels, nor is the padding of a ragged batch's emitted slots: give them
        // defined values (row4col / col4row -1, gain 0)
        e = kb::launch_fill_unused(oNf + b0, nRow ? dNR.as<int32_t>() + b0 : nullptr, nRow ? dNC.as<int32_t>() + b0 : nullptr, nb, k, maxCol, maxRow,
                                   reinterpret_cast<int32_t *>(oR4C + (size_t)b0 * k * maxCol * esz),
                                   oC4R ? reinterpret_cast<int32_t *>(oC4R + (size_t)b0 * k * maxRow * esz) : nullptr, oGain + (size_t)b0 * k, tabI8,
                                   st[c]);
        if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "fill kernel launch", e); break; }
        if (keep && keep->keepI8) {  // the kept table is the byte table (the multi-device exchange travels in bytes)
            e = kb::launch_narrow_i32(reinterpret_cast<const int *>(sR4C) + (size_t)b0 * k * maxCol, keep->row4col8 + (size_t)b0 * k * maxCol,
                                      (long long)nb * k * maxCol, st[c]);
            if (e != hipSuccess) { rc = fail(ctx, KBEST_ERR_HIP, "narrowing kernel launch", e); break; }
        }
        if (nPiece > 1 &&
            (hipEventCreateWithFlags(&done[c], hipEventDisableTiming) != hipSuccess || hipEventRecord(done[c], st[c]) != hipSuccess))
            rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: event", hipGetLastError());
    }
    if (nPiece > 1) {
        for (int c = 0; c < nPiece; c++)  // whatever is ordered behind the context's stream from now on is behind every piece
            if (done[c] && hipStreamWaitEvent(ctx->stream, done[c], 0) != hipSuccess && rc == KBEST_OK)
                rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: wait", hipGetLastError());
        pieceLock.unlock();
    }
    if (rc == KBEST_OK && !direct) {
        for (int c = 0; c < nPiece && rc == KBEST_OK; c++) {
            const int b0 = (int)((long long)B * c / nPiece), nb = (int)((long long)B * (c + 1) / nPiece) - b0;
            hipError_t e = (nPiece > 1) ? hipEventSynchronize(done[c]) : hipStreamSynchronize(ctx->stream);
            const size_t rOff = (size_t)b0 * k * maxCol * esz, cOff = (size_t)b0 * k * maxRow * esz;
            if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<char *>(row4col) + rOff, sR4C + rOff, (size_t)nb * k * maxCol * esz, hipMemcpyDeviceToHost);
            if (e == hipSuccess && col4row) e = hipMemcpy(reinterpret_cast<char *>(col4row) + cOff, sC4R + cOff, (size_t)nb * k * maxRow * esz, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(gain + (size_t)b0 * k, sGain + (size_t)b0 * k, (size_t)nb * k * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(nf + b0, sNf + b0, (size_t)nb * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess && pushed) e = hipMemcpy(pushed + b0, dPushed.as<int64_t>() + b0, (size_t)nb * 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: copy back", e);
        }
    }
    // nothing of this call may still be running when its buffers go back to the cache (and, on the direct path, the tables
    // are complete in the caller's memory only now)
    for (int c = 0; c < nPiece; c++) {
        const hipError_t e = hipStreamSynchronize(st[c]);
        if (e != hipSuccess && rc == KBEST_OK) rc = fail(ctx, KBEST_ERR_HIP, "kbest_batch_f64: synchronize", e);
    }
    for (int c = 0; c < 4; c++)
        if (done[c]) (void)hipEventDestroy(done[c]);
    if (start) (void)hipEventDestroy(start);
    if (rc != KBEST_OK) return rc;
    // (tables written into the caller's registered HOST memory: runs of equal gains were reported, not ordered -- here)
    if (direct && opts->tie_flags && tie_mode(ctx, opts, false)) {
        rc = order_flagged_host(ctx, opts->tie_flags, B, k, maxRow, maxCol, nRow, nCol, gain, row4col, col4row, nf, tabI8);
        if (rc != KBEST_OK) return rc;
    }
    for (int b = 0; b < B; b++)  // shapes were validated above: a negative count can only be an engine failure
        if (nf[b] < 0) return fail(ctx, nf[b] == -1 ? KBEST_ERR_UNSUPPORTED : KBEST_ERR_INTERNAL, "kbest_batch_f64: a problem came back with nf < 0");
    return KBEST_OK;
}

extern "C" {

int kbest_batch_f64(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                    const int32_t *nRow, const int32_t *nCol, const double *cost,
                    const int64_t *costOff, int k, int32_t *row4col, int32_t *col4row, double *gain,
                    int32_t *nf, int64_t *pushed)
{
    if (!ctx || !opts || B <= 0 || k < 1)
        return kbest_batch_f64_keep(ctx, opts, B, maxRow, maxCol, nRow, nCol, cost, costOff, k, row4col, col4row, gain, nf, pushed, nullptr);
    if (!tie_mode(ctx, opts, false)) {
        // no tie check in this mode (the reference's own order, push counting, ...): nothing to flag -- and this entry's tie_flags is a
        // HOST array, which must not travel down as the device pointer the launches take
        kbest_opts o = *opts;
        if (o.tie_flags) memset(o.tie_flags, 0, (size_t)B * 4);
        o.tie_flags = nullptr;
        return kbest_batch_f64_keep(ctx, &o, B, maxRow, maxCol, nRow, nCol, cost, costOff, k, row4col, col4row, gain, nf, pushed, nullptr);
    }
    // Exact ties (kbest_ties.h; "Order of exact ties" in kbest_c.h).  The launch reports per problem whether any two of its k + 1
    // best gains are equal; this synchronous entry then gives those problems the REFERENCE's answer -- they run again on the
    // reference-order kernel -- or, with KBEST_FLAG_CANONICAL_TIES, completes a level that straddles slot k under the engine's own rule
    // (kb_complete_tie_levels): either way the answer does not depend on the kernel the batch was routed to.
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dFlags;
    HIP_TRY(ctx, dFlags.alloc(ctx, (size_t)B * 4));
    kbest_opts o = *opts;
    o.tie_flags = dFlags.as<int32_t>();
    int rc = kbest_batch_f64_keep(ctx, &o, B, maxRow, maxCol, nRow, nCol, cost, costOff, k, row4col, col4row, gain, nf, pushed, nullptr);
    if (rc != KBEST_OK) return rc;
    std::vector<int32_t> fl((size_t)B);
    HIP_TRY(ctx, hipMemcpy(fl.data(), dFlags.as<int32_t>(), (size_t)B * 4, hipMemcpyDeviceToHost));
    if (!(opts->flags & KBEST_FLAG_NO_TIE_RESOLVE))
        kb_complete_tie_levels(ctx, opts, B, maxRow, maxCol, nRow, nCol, cost, costOff, k, row4col, col4row, gain, fl.data(), nullptr);
    for (int b = 0; b < B; b++) {
        if ((fl[b] & KBEST_TIE_BOUNDARY) && !(fl[b] & KBEST_TIE_RESOLVED)) fl[b] |= KBEST_TIE_UNRESOLVED;
        if ((fl[b] & KBEST_TIE_UNORDERED) && !(fl[b] & KBEST_TIE_RESOLVED)) {
            // a run of more than 4 096 equal gains inside the first pass' tables: the tables are in the caller's memory -- ordered here
            const bool i8 = (opts->flags & KBEST_FLAG_TABLES_I8) != 0;
            const int n = nf[b] < 0 ? 0 : (nf[b] > k ? k : nf[b]), M = nCol ? nCol[b] : maxCol, N = nRow ? nRow[b] : maxRow;
            if (i8)
                order_ties_host<signed char>(gain + (size_t)b * k, reinterpret_cast<signed char *>(row4col) + (size_t)b * k * maxCol,
                                             col4row ? reinterpret_cast<signed char *>(col4row) + (size_t)b * k * maxRow : nullptr, n, M, N, maxCol, maxRow);
            else
                order_ties_host<int32_t>(gain + (size_t)b * k, row4col + (size_t)b * k * maxCol, col4row ? col4row + (size_t)b * k * maxRow : nullptr, n, M, N,
                                         maxCol, maxRow);
        }
        fl[b] &= ~KBEST_TIE_UNORDERED;
    }
    if (opts->tie_flags) memcpy(opts->tie_flags, fl.data(), (size_t)B * 4);
    {
        std::lock_guard<std::mutex> lock(ctx->tieMu);
        ctx->lastTie.swap(fl);
    }
    return KBEST_OK;
}

}  // extern "C"

// What the synchronous entries do with the problems a launch flagged.  By default: every problem with an exact tie among its k + 1
// best gains again on the reference-order kernel, its tables replaced (KBEST_TIE_REFERENCE; the first block below).  With
// KBEST_FLAG_CANONICAL_TIES (or where that re-run fails): completes the gain levels that straddle slot k under the engine's own rule
// (fl[b] & KBEST_TIE_BOUNDARY without KBEST_TIE_RESOLVED): those problems again,
// alone, with k + 64, then k + 256, k + 1 024, then k + KBEST_TIE_CAP solutions -- whichever kernel takes that k; the table comes back in the
// canonical order -- until the level ends inside the table; the first k of the ordered table then replace the problem's slots in the
// caller's HOST tables (row4col / col4row: int32, or int8 with KBEST_FLAG_TABLES_I8; col4row may be null) and the problem is flagged
// KBEST_TIE_RESOLVED.  Nothing of a problem is touched before its re-run has validated; a re-run that fails (beyond a kernel's
// limits, out of memory) or a level of more than KBEST_TIE_CAP members beyond k leaves the first pass' tables and the flags as they
// are (the caller marks such a problem KBEST_TIE_UNRESOLVED).  changed (optional): the problems whose tables were replaced.
// cost / costOff / nRow / nCol: as kbest_batch_f64's (host).
void kb_complete_tie_levels(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow, const int32_t *nCol,
                            const double *cost, const int64_t *costOff, int k, void *row4col, void *col4row, double *gain, int32_t *fl,
                            std::vector<int> *changed)
{
    static const int steps[4] = {64, 256, 1024, KBEST_TIE_CAP};
    const bool i8 = (opts->flags & KBEST_FLAG_TABLES_I8) != 0;
    const size_t esz = i8 ? 1 : 4;
    if (!(opts->flags & KBEST_FLAG_CANONICAL_TIES)) {
        // The REFERENCE's answer wherever gains tie (kbest_c.h; the default of the synchronous entries): every problem whose k + 1 best
        // gains hold an exact tie -- inside the table or across slot k -- or that could not be checked is enumerated again by the
        // reference-order kernel (kbest_exact.hip), and its tables replace the first pass'.  Tie-free problems keep the fast kernels'
        // tables: those ARE the reference's.  (KBEST_FLAG_CANONICAL_TIES: the engine's own rule instead -- the steps below.)
        const int tied = KBEST_TIE_INSIDE | KBEST_TIE_BOUNDARY | KBEST_TIE_UNCHECKED | KBEST_TIE_UNORDERED;
        std::vector<int> idx;
        for (int b = 0; b < B; b++)
            if ((fl[b] & tied) && !(fl[b] & KBEST_TIE_REFERENCE)) idx.push_back(b);
        if (idx.empty()) return;
        const int n = (int)idx.size();
        std::vector<int32_t> sRow(n), sCol(n), sNf(n);
        std::vector<int64_t> sOff(n);
        size_t tot = 0;
        for (int i = 0; i < n; i++) {
            sRow[i] = nRow ? nRow[idx[i]] : maxRow;
            sCol[i] = nCol ? nCol[idx[i]] : maxCol;
            sOff[i] = (int64_t)tot;
            tot += (size_t)sRow[i] * sCol[i];
        }
        std::vector<double> sCost(tot), sGain((size_t)n * k);
        for (int i = 0; i < n; i++) {
            const size_t src = costOff ? (size_t)costOff[idx[i]] : (size_t)idx[i] * maxRow * maxCol;
            memcpy(sCost.data() + sOff[i], cost + src, (size_t)sRow[i] * sCol[i] * 8);
        }
        std::vector<char> sR((size_t)n * k * maxCol * esz), sC(col4row ? (size_t)n * k * maxRow * esz : 0);
        kbest_opts o2 = *opts;
        o2.flags = (o2.flags & ~KBEST_FLAG_REFERENCE_TIES) | KBEST_FLAG_REFERENCE_ORDER;
        o2.tie_flags = nullptr;
        const int rc = kbest_batch_f64_keep(ctx, &o2, n, maxRow, maxCol, sRow.data(), sCol.data(), sCost.data(), sOff.data(), k,
                                            reinterpret_cast<int32_t *>(sR.data()), col4row ? reinterpret_cast<int32_t *>(sC.data()) : nullptr, sGain.data(),
                                            sNf.data(), nullptr, nullptr);
        // (a re-run that fails -- no memory for a pool of hypotheses -- leaves the first pass' tables, in the engine's own order, and
        //  the engine's own rule completes the levels at slot k below)
        for (int i = 0; i < n && rc == KBEST_OK; i++) {
            const int b = idx[i];
            memcpy(static_cast<char *>(row4col) + (size_t)b * k * maxCol * esz, sR.data() + (size_t)i * k * maxCol * esz, (size_t)k * maxCol * esz);
            if (col4row)
                memcpy(static_cast<char *>(col4row) + (size_t)b * k * maxRow * esz, sC.data() + (size_t)i * k * maxRow * esz, (size_t)k * maxRow * esz);
            memcpy(gain + (size_t)b * k, sGain.data() + (size_t)i * k, (size_t)k * 8);
            fl[b] = (fl[b] & KBEST_TIE_INSIDE) | KBEST_TIE_REFERENCE;
            if (changed) changed->push_back(b);
        }
        if (rc == KBEST_OK) return;
    }
    for (int step = 0; step < 4; step++) {
        std::vector<int> idx;
        for (int b = 0; b < B; b++)
            if ((fl[b] & KBEST_TIE_BOUNDARY) && !(fl[b] & KBEST_TIE_RESOLVED)) idx.push_back(b);
        if (idx.empty()) return;
        const int n = (int)idx.size(), k2 = k + steps[step];
        std::vector<int32_t> sRow(n), sCol(n), sNf(n);
        std::vector<int64_t> sOff(n);
        size_t tot = 0;
        for (int i = 0; i < n; i++) {
            sRow[i] = nRow ? nRow[idx[i]] : maxRow;
            sCol[i] = nCol ? nCol[idx[i]] : maxCol;
            sOff[i] = (int64_t)tot;
            tot += (size_t)sRow[i] * sCol[i];
        }
        std::vector<double> sCost(tot), sGain((size_t)n * k2);
        for (int i = 0; i < n; i++) {
            const size_t src = costOff ? (size_t)costOff[idx[i]] : (size_t)idx[i] * maxRow * maxCol;
            memcpy(sCost.data() + sOff[i], cost + src, (size_t)sRow[i] * sCol[i] * 8);
        }
        std::vector<char> sR((size_t)n * k2 * maxCol * esz), sC(col4row ? (size_t)n * k2 * maxRow * esz : 0);
        kbest_opts o2 = *opts;
        DevBuf dFlags2;
        if (dFlags2.alloc(ctx, (size_t)n * 4) != hipSuccess) return;
        o2.tie_flags = dFlags2.as<int32_t>();
        const int rc = kbest_batch_f64_keep(ctx, &o2, n, maxRow, maxCol, sRow.data(), sCol.data(), sCost.data(), sOff.data(), k2,
                                            reinterpret_cast<int32_t *>(sR.data()), col4row ? reinterpret_cast<int32_t *>(sC.data()) : nullptr, sGain.data(),
                                            sNf.data(), nullptr, nullptr);
        if (rc != KBEST_OK) return;  // (the first pass' tables stand; the problems stay flagged)
        std::vector<int32_t> fl2((size_t)n);
        if (hipMemcpy(fl2.data(), dFlags2.as<int32_t>(), (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
        for (int i = 0; i < n; i++) {
            const int b = idx[i];
            const double *g2 = sGain.data() + (size_t)i * k2;
            // the level is complete when the table goes on beyond it (or the problem has no more assignments) -- and the table is
            // in the canonical order throughout (a run of more than 4 096 equal gains is left as the kernel emitted it)
            const bool complete = sNf[i] >= k && (sNf[i] < k2 || g2[k2 - 1] != g2[k - 1]);
            if (!complete) continue;
            if (fl2[i] & KBEST_TIE_UNORDERED) {
                // a run of more than 4 096 equal gains, which the finishing launch leaves as the kernel emitted it: ordered here (the
                // re-run's tables are in host memory)
                const int nn = sNf[i] < k2 ? sNf[i] : k2;
                if (i8)
                    order_ties_host<signed char>(g2, reinterpret_cast<signed char *>(sR.data()) + (size_t)i * k2 * maxCol,
                                                 col4row ? reinterpret_cast<signed char *>(sC.data()) + (size_t)i * k2 * maxRow : nullptr, nn, sCol[i], sRow[i], maxCol, maxRow);
                else
                    order_ties_host<int32_t>(g2, reinterpret_cast<int32_t *>(sR.data()) + (size_t)i * k2 * maxCol,
                                             col4row ? reinterpret_cast<int32_t *>(sC.data()) + (size_t)i * k2 * maxRow : nullptr, nn, sCol[i], sRow[i], maxCol, maxRow);
            }
            memcpy(static_cast<char *>(row4col) + (size_t)b * k * maxCol * esz, sR.data() + (size_t)i * k2 * maxCol * esz, (size_t)k * maxCol * esz);
            if (col4row)
                memcpy(static_cast<char *>(col4row) + (size_t)b * k * maxRow * esz, sC.data() + (size_t)i * k2 * maxRow * esz, (size_t)k * maxRow * esz);
            memcpy(gain + (size_t)b * k, g2, (size_t)k * 8);
            fl[b] |= KBEST_TIE_RESOLVED;
            if (changed) changed->push_back(b);
        }
    }
}

extern "C" {

// The asynchronous entries report a tie at slot k, they cannot complete it.  This SYNCHRONOUS helper does, for the tables of an
// earlier kbest_batch_f64_dev call that are still on the device: it waits for `stream`, reads the flags, completes the flagged gain
// levels (kb_complete_tie_levels: the flagged problems again with k + 64 / 256 / 1 024 solutions) and patches those problems' slots
// of the device tables and flags in place.
int kbest_resolve_ties_dev(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *d_nRow,
                           const int32_t *d_nCol, const double *d_cost, const int64_t *d_costOff, int k, int32_t *d_row4col,
                           int32_t *d_col4row, double *d_gain, int32_t *d_tie_flags, void *stream)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (!opts || B < 0 || k < 1 || maxCol < 1 || maxRow < maxCol || !d_cost || !d_row4col || !d_gain || !d_tie_flags ||
        (d_nRow == nullptr) != (d_nCol == nullptr) || (d_costOff && !d_nRow))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_resolve_ties_dev: bad argument");
    if (B == 0) return KBEST_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : ctx->stream));
    std::vector<int32_t> fl((size_t)B);
    HIP_TRY(ctx, hipMemcpy(fl.data(), d_tie_flags, (size_t)B * 4, hipMemcpyDeviceToHost));
    std::vector<int> idx;
    const bool refTies = !(opts->flags & KBEST_FLAG_CANONICAL_TIES);
    const int tied = KBEST_TIE_INSIDE | KBEST_TIE_BOUNDARY | KBEST_TIE_UNCHECKED | KBEST_TIE_UNORDERED;
    for (int b = 0; b < B; b++)
        if (refTies ? ((fl[b] & tied) && !(fl[b] & KBEST_TIE_REFERENCE)) : ((fl[b] & KBEST_TIE_BOUNDARY) && !(fl[b] & KBEST_TIE_RESOLVED))) idx.push_back(b);
    if (idx.empty()) return KBEST_OK;
    // the flagged problems' cost blocks and table slots come to the host, are completed there, and go back
    const int n = (int)idx.size();
    const bool i8 = (opts->flags & KBEST_FLAG_TABLES_I8) != 0;
    const size_t esz = i8 ? 1 : 4, per = (size_t)maxRow * maxCol;
    std::vector<int32_t> hR, hC, sRow(n), sCol(n), sFl(n);
    std::vector<int64_t> hOff, sOff(n);
    if (d_nRow) {
        hR.resize(B); hC.resize(B);
        HIP_TRY(ctx, hipMemcpy(hR.data(), d_nRow, (size_t)B * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(hC.data(), d_nCol, (size_t)B * 4, hipMemcpyDeviceToHost));
    }
    if (d_costOff) {
        hOff.resize(B);
        HIP_TRY(ctx, hipMemcpy(hOff.data(), d_costOff, (size_t)B * 8, hipMemcpyDeviceToHost));
    }
    std::vector<double> sCost((size_t)n * per), sGain((size_t)n * k);
    std::vector<char> sR((size_t)n * k * maxCol * esz), sC(d_col4row ? (size_t)n * k * maxRow * esz : 0);
    for (int i = 0; i < n; i++) {
        const int b = idx[i];
        sRow[i] = d_nRow ? hR[b] : maxRow;
        sCol[i] = d_nRow ? hC[b] : maxCol;
        sOff[i] = (int64_t)i * (int64_t)per;
        sFl[i] = fl[b];
        const size_t src = d_costOff ? (size_t)hOff[b] : (size_t)b * per;
        HIP_TRY(ctx, hipMemcpy(sCost.data() + (size_t)i * per, d_cost + src, (size_t)sRow[i] * sCol[i] * 8, hipMemcpyDeviceToHost));
    }
    std::vector<int> changed;
    kb_complete_tie_levels(ctx, opts, n, maxRow, maxCol, sRow.data(), sCol.data(), sCost.data(), sOff.data(), k, sR.data(), d_col4row ? sC.data() : nullptr,
                           sGain.data(), sFl.data(), &changed);
    for (int i : changed) {
        const int b = idx[i];
        HIP_TRY(ctx, hipMemcpy(reinterpret_cast<char *>(d_row4col) + (size_t)b * k * maxCol * esz, sR.data() + (size_t)i * k * maxCol * esz, (size_t)k * maxCol * esz, hipMemcpyHostToDevice));
        if (d_col4row)
            HIP_TRY(ctx, hipMemcpy(reinterpret_cast<char *>(d_col4row) + (size_t)b * k * maxRow * esz, sC.data() + (size_t)i * k * maxRow * esz, (size_t)k * maxRow * esz, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(d_gain + (size_t)b * k, sGain.data() + (size_t)i * k, (size_t)k * 8, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < n; i++) {
        int32_t f = sFl[i];
        if ((f & KBEST_TIE_BOUNDARY) && !(f & KBEST_TIE_RESOLVED)) f |= KBEST_TIE_UNRESOLVED;
        HIP_TRY(ctx, hipMemcpy(d_tie_flags + idx[i], &f, 4, hipMemcpyHostToDevice));
    }
    return KBEST_OK;
}

static int merge_topk_impl(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const void *d_gain,
                           const void *d_row4col, const void *d_nf, int64_t shardStrideBytes, double *d_outGain,
                           int32_t *d_outRow4col, int32_t *d_outNf, void *stream, bool inI8)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || nShard < 1 || k < 1 || maxCol < 1 || !d_gain || !d_row4col || !d_nf || !d_outGain || !d_outRow4col || !d_outNf ||
        (nShard > 1 && shardStrideBytes <= 0))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_merge_topk_f64_dev: bad argument");
    if (B == 0) return KBEST_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kb::MergeParams p;
    p.gain = static_cast<const unsigned char *>(d_gain);
    p.row4col = static_cast<const unsigned char *>(d_row4col);
    p.nf = static_cast<const unsigned char *>(d_nf);
    p.shardStride = shardStrideBytes;
    p.nShard = nShard;
    p.k = k;
    p.maxCol = maxCol;
    p.ldCol = maxCol;
    p.maximize = maximize;
    p.outGain = d_outGain;
    p.outRow4col = d_outRow4col;
    p.outNf = d_outNf;
    p.outCol4row = nullptr;
    p.ldRow = 0;
    p.strideR4C = 0;
    p.strideNf = 0;
    p.inI8 = inI8 ? 1 : 0;
    p.spd = 1;
    hipError_t e = kb::launch_merge_topk(p, B, stream ? static_cast<hipStream_t>(stream) : ctx->stream);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "merge kernel launch", e);
    return KBEST_OK;
}

int kbest_merge_topk_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const void *d_gain,
                             const void *d_row4col, const void *d_nf, int64_t shardStrideBytes, double *d_outGain,
                             int32_t *d_outRow4col, int32_t *d_outNf, void *stream)
{
    return merge_topk_impl(ctx, B, nShard, k, maxCol, maximize, d_gain, d_row4col, d_nf, shardStrideBytes, d_outGain, d_outRow4col, d_outNf, stream, false);
}

int kbest_merge_topk_i8_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const void *d_gain,
                                const void *d_row4col8, const void *d_nf, int64_t shardStrideBytes, double *d_outGain,
                                int32_t *d_outRow4col, int32_t *d_outNf, void *stream)
{
    return merge_topk_impl(ctx, B, nShard, k, maxCol, maximize, d_gain, d_row4col8, d_nf, shardStrideBytes, d_outGain, d_outRow4col, d_outNf, stream, true);
}

int kbest_merge_gains_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const double *d_gain, const int32_t *d_nf,
                              int ownShard, const int8_t *d_ownRow4col8, double *d_outGain, int8_t *d_outRow4col8, int32_t *d_outNf,
                              int32_t *d_tied, void *stream)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || nShard < 1 || k < 1 || maxCol < 1 || !d_gain || !d_nf || ownShard < 0 || ownShard >= nShard || !d_ownRow4col8 || !d_outGain ||
        !d_outRow4col8 || !d_outNf || !d_tied)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_merge_gains_f64_dev: bad argument");
    if (B == 0) return KBEST_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kb::MergeGainsParams q;
    memset(&q, 0, sizeof(q));
    q.gain = reinterpret_cast<const unsigned char *>(d_gain);
    q.nf = reinterpret_cast<const unsigned char *>(d_nf);
    q.blockStride = 0;
    q.spd = nShard;  // plain [nShard][B][...] arrays: one block
    q.ownRow8 = reinterpret_cast<const signed char *>(d_ownRow4col8);
    q.ownLo = ownShard;
    q.ownHi = ownShard + 1;
    q.nShard = nShard;
    q.B = B;
    q.k = k;
    q.maxCol = maxCol;
    q.maximize = maximize;
    q.outGain = d_outGain;
    q.outRow8 = reinterpret_cast<signed char *>(d_outRow4col8);
    q.outNf = d_outNf;
    q.tied = d_tied;
    hipError_t e = kb::launch_merge_gains(q, stream ? static_cast<hipStream_t>(stream) : ctx->stream);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "merge kernel launch", e);
    return KBEST_OK;
}

int kbest_assign_batch_f64(kbest_ctx *ctx, int B, int maxRow, int maxCol, const int32_t *nRow, const int32_t *nCol,
                           const double *cost, const int64_t *costOff, int maximize, int shift, int gainCols,
                           int32_t *row4col, int32_t *col4row, double *gain, double *u, double *v, int32_t *feasible)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || maxCol < 1 || maxRow < maxCol || !cost || !row4col || !col4row || !gain || !feasible || gainCols < 0 ||
        (!shift && maximize))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_assign_batch_f64: bad argument");
    if ((nRow == nullptr) != (nCol == nullptr) || (costOff && !nRow))
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_assign_batch_f64: give nRow, nCol (and costOff) together");
    if (maxRow > KBEST_MAX_DIM) return fail(ctx, KBEST_ERR_UNSUPPORTED, "kbest_assign_batch_f64: numRow > KBEST_MAX_DIM");
    if (B == 0) return KBEST_OK;
    size_t nCost = (size_t)B * maxRow * maxCol;
    if (nRow) {
        nCost = 0;
        for (int b = 0; b < B; b++) {
            if (nCol[b] < 1 || nRow[b] < nCol[b] || nRow[b] > maxRow || nCol[b] > maxCol)
                return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_assign_batch_f64: shape out of range (need 1 <= numCol <= numRow <= maxRow)");
            const size_t end = (costOff ? (size_t)costOff[b] : (size_t)b * maxRow * maxCol) + (size_t)nRow[b] * nCol[b];
            if (end > nCost) nCost = end;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dCost, dMeta, dOut;
    // one block in (shapes, offsets), one block out (row4col, col4row, nf | gain, u, v)
    const size_t B4 = ((size_t)B * 4 + 15) & ~(size_t)15, B8 = ((size_t)B * 8 + 15) & ~(size_t)15;
    std::vector<unsigned char> meta(2 * B4 + B8);
    if (nRow) { memcpy(meta.data(), nRow, (size_t)B * 4); memcpy(meta.data() + B4, nCol, (size_t)B * 4); }
    if (costOff) memcpy(meta.data() + 2 * B4, costOff, (size_t)B * 8);
    const size_t oR4C = 0, oC4R = oR4C + (((size_t)B * maxCol * 4 + 15) & ~(size_t)15),
                 oNf = oC4R + (((size_t)B * maxRow * 4 + 15) & ~(size_t)15), oGain = oNf + B4, oU = oGain + B8,
                 oV = oU + (size_t)B * maxCol * 8, oEnd = oV + (size_t)B * maxRow * 8;
    HIP_TRY(ctx, dCost.alloc(ctx, nCost * 8));
    HIP_TRY(ctx, dMeta.alloc(ctx, meta.size()));
    HIP_TRY(ctx, dOut.alloc(ctx, oEnd));
    HIP_TRY(ctx, hipMemcpy(dCost.p, cost, nCost * 8, hipMemcpyHostToDevice));
    if (nRow) HIP_TRY(ctx, hipMemcpy(dMeta.p, meta.data(), meta.size(), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemsetAsync(dOut.p, 0xFF, oGain, ctx->stream));          // indices of infeasible problems: -1
    HIP_TRY(ctx, hipMemsetAsync(dOut.as<unsigned char>() + oGain, 0, oEnd - oGain, ctx->stream));
    unsigned char *ob = dOut.as<unsigned char>(), *mb = dMeta.as<unsigned char>();
    kbest_opts o;
    kbest_default_opts(&o);
    o.maximize = maximize ? 1 : 0;
    o.flags = KBEST_FLAG_RECT_ROOT | (shift ? 0u : KBEST_FLAG_NO_SHIFT);
    DevExtra ex;
    ex.dualU = reinterpret_cast<double *>(ob + oU);
    ex.dualV = reinterpret_cast<double *>(ob + oV);
    ex.gainCols = gainCols;
    int rc = batch_dev_impl(ctx, &o, B, maxRow, maxCol, nRow ? reinterpret_cast<int32_t *>(mb) : nullptr,
                            nRow ? reinterpret_cast<int32_t *>(mb + B4) : nullptr, dCost.as<double>(),
                            costOff ? reinterpret_cast<int64_t *>(mb + 2 * B4) : nullptr, 1,
                            reinterpret_cast<int32_t *>(ob + oR4C), reinterpret_cast<int32_t *>(ob + oC4R),
                            reinterpret_cast<double *>(ob + oGain), reinterpret_cast<int32_t *>(ob + oNf), nullptr,
                            ctx->stream, true, &ex);
    if (rc != KBEST_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned char> out(oEnd);
    HIP_TRY(ctx, hipMemcpy(out.data(), dOut.p, oEnd, hipMemcpyDeviceToHost));
    memcpy(row4col, out.data() + oR4C, (size_t)B * maxCol * 4);
    memcpy(col4row, out.data() + oC4R, (size_t)B * maxRow * 4);
    memcpy(feasible, out.data() + oNf, (size_t)B * 4);
    memcpy(gain, out.data() + oGain, (size_t)B * 8);
    if (u) memcpy(u, out.data() + oU, (size_t)B * maxCol * 8);
    if (v) memcpy(v, out.data() + oV, (size_t)B * maxRow * 8);
    for (int b = 0; b < B; b++) {
        if (feasible[b] < 0) return fail(ctx, KBEST_ERR_INTERNAL, "kbest_assign_batch_f64: a problem came back with nf < 0");
        if (feasible[b] == 0) gain[b] = -1.0;  // shortestPathCPP's infeasibility marker (cpp:200)
    }
    return KBEST_OK;
}

int kbest_to_probs_f64(kbest_ctx *ctx, double *x, int64_t n)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (n < 0 || (n > 0 && !x)) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_to_probs_f64: bad argument");
    if (n == 0) return KBEST_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIP_TRY(ctx, d.alloc(ctx, (size_t)n * 8));
    HIP_TRY(ctx, hipMemcpy(d.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
    hipError_t e = kb::launch_to_probs(d.as<double>(), (long long)n, ctx->stream);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "toProbs kernel launch", e);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(x, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return KBEST_OK;
}

// Shared body of kbest_weights_batch_f64 (condition = false: assignmentProb on the given matrices) and
// kbest_assoc_probs_batch_f64 (condition = true: conditionCosts -> assignmentProb -> scatter back, i.e.
// getAssignmentProbs assignment.cpp:57-74 without the quadric cost construction).  Everything between the
// H2D copy of the cost blocks and the D2H copy of the probabilities runs on the device, stream-ordered.
struct QuadricHost {  // host-side inputs of computeQuadricCostMatrix, packed frame after frame
    const double *landMean, *landCov, *measMean, *measCov;
    double gate;
};

static int arena_reserve(kbest_ctx *ctx, kbest_ctx::Arena &a, size_t need)
{
    if (need <= a.bytes) return KBEST_OK;
    size_t cap = 1 << 16;
    while (cap < need) cap <<= 1;
    if (a.host) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(a.host); a.host = nullptr; a.bytes = 0; }
    hipError_t e = hipHostMalloc(&a.host, cap, hipHostMallocMapped);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_NOMEM, "hipHostMalloc(staging)", e);
    e = hipHostGetDevicePointer(&a.dev, a.host, 0);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "hipHostGetDevicePointer", e);
    a.bytes = cap;
    return KBEST_OK;
}

static int raw_reserve(kbest_ctx *ctx, DevBufRaw &d, size_t need)
{
    if (need <= d.bytes) return KBEST_OK;
    size_t cap = 1 << 16;
    while (cap < need) cap <<= 1;
    if (d.p) { HIP_TRY(ctx, hipDeviceSynchronize()); (void)hipFree(d.p); d.p = nullptr; d.bytes = 0; }  // (rare: only when it grows)
    hipError_t e = hipMalloc(&d.p, cap);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_NOMEM, "hipMalloc(staging)", e);
    d.bytes = cap;
    return KBEST_OK;
}

// Frames with a handful of measurements -- the reference's real ones (README.md:11: 3-5 per frame) -- have so few assignments in
// all, (nL + nM)! / nL! before conditioning, that looking at every one of them beats enumerating the k best (kbest_tiny.hip).  The
// whole batch must qualify (one launch); assignmentProb's mode only (cutoff 42, gate), on raw blocks (conditionCosts first) or on
// conditioned ones.
static bool tiny_takes(const kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, int k, bool condition, bool bruteForce)
{
    (void)condition;  // (without conditioning: assignmentProb on a block that IS conditioned -- the kernel checks, -2 otherwise)
    if (ctx->noTiny || bruteForce || k > kb::SMALL_MAX_K) return false;
    if (kb::tiny_lds_bytes(k) > ctx->ldsLimit) return false;
    for (int b = 0; b < B; b++) {
        const int m = nM[b], l = nL[b];
        if (m == 0) continue;  // (an empty frame: answered by either kernel at its shape test)
        if (m < 2 || m > kb::TINY_MAX_COL || l < 0 || l + m > kb::TINY_MAX_ROW) return false;
        long long cnt = 1, pre = 1;
        for (int c = 0; c < m; c++) {
            cnt *= (l + m - c);
            if (c < m - 2) pre *= (l + m - c);
        }
        if (cnt > kb::TINY_MAX_COUNT || pre > kb::TINY_MAX_PREFIX) return false;
    }
    return true;
}

static bool bnb_many(const kbest_ctx *ctx, int B) { return B >= (ctx->bnbSmallFrom >= 0 ? ctx->bnbSmallFrom : ctx->nCU + 1); }

// Frame-sized blocks of up to 16 measurements and 64 rows: the bounded walk (kbest_bnb.hip) finds the k best without
// enumerating -- every assignment below a bound that is raised until k lie below it.  assignmentProb's mode only, whole batch.
static bool bnb_takes(const kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, int k, bool bruteForce)
{
    const int nThreads = bnb_many(ctx, B) ? 256 : 1024;  // (launch_kbest_bnb: a batch that fills the chip takes the small workgroups)
    if (ctx->noBnb || bruteForce || k > kb::SMALL_MAX_K || k > kb::bnb_max_k(nThreads)) return false;
    if (kb::bnb_lds_bytes(k, nThreads) > ctx->ldsLimit) return false;
    for (int b = 0; b < B; b++) {
        const int m = nM[b], l = nL[b];
        if (m == 0) continue;
        if (m < 2 || m > kb::BNB_MAX_COL || l < 0 || l + m > kb::BNB_MAX_ROW) return false;
    }
    return true;
}

// The association path on the small-problem kernel (kbest_small.hip): conditionCosts -> kBest2DCutoff(42) -> weights
// -> scatter back, ONE launch, only [nM][nL+1] doubles per frame come back.  Returns 1 when some frame does not fit
// that kernel (more than 32 kept rows, ...): the caller then runs the general pipeline.
struct QuadricHost;
static int weights_small(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const int32_t *nRow, const double *cost,
                         const double *d_cost, const int64_t *costOff, int k, double *probs, const int64_t *probOff,
                         int32_t *nf, bool condition, bool bruteForce, int rawMaxRow, int maxCol, size_t nCost, size_t nProb,
                         std::vector<int> *unfit = nullptr, bool allowFast = true, int32_t *tie = nullptr)
{
    // exact ties (kbest_ties.h): the enumeration kernel enumerates one solution more than the k it weighs and flags a frame
    // whose k-th and (k+1)-th gains are equal (tie[b], KBEST_TIE_BOUNDARY: the caller completes that gain level through the
    // general pipeline); the exhaustive kernel and the bounded walk see the whole level and keep its lexicographically first
    // assignments themselves (KBEST_TIE_BOUNDARY | KBEST_TIE_RESOLVED)
    const bool tieOn = !ctx->noTie;
    const int kT = k;
    const int capRow = rawMaxRow < kb::SMALL_MAX_DIM ? rawMaxRow : kb::SMALL_MAX_DIM;
    int nw = 0;
    // (a synchronous entry: where k + 1 no longer fits this kernel the general pipeline, which does take it, answers -- see batch_dev_impl)
    const bool extraSol = tieOn;
    if (extraSol) k = k + 1;
    if (!small_fits(ctx, B, capRow, maxCol, k, true, &nw)) return 1;
    if (!condition && rawMaxRow > kb::SMALL_MAX_DIM) return 1;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        int rc = order_behind_last(ctx, ctx->stream);
        if (rc != KBEST_OK) return rc;
        rc = ensure_states(ctx, small_states_need(B, capRow, maxCol, k, nw), true);
        if (rc != KBEST_OK) return rc;
    }
    // in: [costOff | probOff | nRow | nM | nL] [cost blocks]      out: [probabilities] [nf]
    const size_t B8 = ((size_t)B * 8 + 15) & ~(size_t)15, B4 = ((size_t)B * 4 + 15) & ~(size_t)15;
    const size_t metaBytes = 2 * B8 + 3 * B4;
    // cost blocks / probabilities in registered caller memory (kbest_register_host_buffer) are used where they lie
    if (!d_cost) d_cost = static_cast<const double *>(mapped(ctx, cost, nCost * 8));
    double *mProbs = static_cast<double *>(mapped(ctx, probs, nProb * 8));
    const size_t inBytes = metaBytes + (d_cost ? 0 : nCost * 8);
    const size_t probBytes = (nProb * 8 + 15) & ~(size_t)15;
    const size_t nfBytes = ((size_t)B * 4 + 15) & ~(size_t)15;
    const size_t outBytes = probBytes + 2 * nfBytes;  // probabilities | nf | tie flags
    {
        int rc = arena_reserve(ctx, ctx->pinIn, inBytes);
        if (rc == KBEST_OK) rc = arena_reserve(ctx, ctx->pinOut, outBytes + 64);  // (+ the completion counter of zero-copy calls)
        if (rc != KBEST_OK) return rc;
    }
    unsigned char *hin = static_cast<unsigned char *>(ctx->pinIn.host);
    memcpy(hin, costOff, (size_t)B * 8);
    memcpy(hin + B8, probOff, (size_t)B * 8);
    memcpy(hin + 2 * B8, nRow, (size_t)B * 4);
    memcpy(hin + 2 * B8 + B4, nM, (size_t)B * 4);
    memcpy(hin + 2 * B8 + 2 * B4, nL, (size_t)B * 4);
    if (!d_cost) memcpy(hin + metaBytes, cost, nCost * 8);
    // The kernel reads every cost block once and writes every probability once: it works on the pinned (device-mapped)
    // staging memory itself, no copy engine round trips (1 000 frames per call: 0.79 -> 0.74 ms, 4 000: 2.39 -> 2.12).
    const bool zeroCopy = inBytes + outBytes <= ctx->zcLimit;
    if (!zeroCopy) mProbs = nullptr;
    unsigned char *din, *dout;
    if (zeroCopy) {
        din = static_cast<unsigned char *>(ctx->pinIn.dev);
        dout = static_cast<unsigned char *>(ctx->pinOut.dev);
    } else {
        int rc = raw_reserve(ctx, ctx->stageIn, inBytes);
        if (rc == KBEST_OK) rc = raw_reserve(ctx, ctx->stageOut, outBytes);
        if (rc != KBEST_OK) return rc;
        din = static_cast<unsigned char *>(ctx->stageIn.p);
        dout = static_cast<unsigned char *>(ctx->stageOut.p);
        HIP_TRY(ctx, hipMemcpyAsync(din, hin, inBytes, hipMemcpyHostToDevice, ctx->stream));
    }
    kb::SmallParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.cost = d_cost ? d_cost : reinterpret_cast<const double *>(din + metaBytes);
    sp.costOff = reinterpret_cast<const long long *>(din);
    sp.probOff = reinterpret_cast<const long long *>(din + B8);
    sp.nRow = reinterpret_cast<const int *>(din + 2 * B8);
    sp.nCol = reinterpret_cast<const int *>(din + 2 * B8 + B4);
    sp.nL = reinterpret_cast<const int *>(din + 2 * B8 + 2 * B4);
    sp.maxRow = capRow;
    sp.maxCol = maxCol;
    sp.bnbRow = rawMaxRow < kb::BNB_MAX_ROW ? rawMaxRow : kb::BNB_MAX_ROW;
    sp.ldRow = capRow;
    sp.ldCol = maxCol;
    sp.k = k;
    sp.kTab = kT;
    sp.maximize = 0;
    sp.useCutoff = bruteForce ? 0 : 1;  // assignment.cpp:594: kBest2DCutoff(..., cutoff = 42); :880: plain kBest2D
    sp.cutoff = 42.0;
    sp.nf = reinterpret_cast<int *>(dout + probBytes);
    sp.tieFlags = tieOn ? reinterpret_cast<int *>(dout + probBytes + nfBytes) : nullptr;
    sp.tieBase = (tieOn && !extraSol) ? KBEST_TIE_UNCHECKED : 0;
    sp.states = ctx->states;
    sp.stateStride = kb::small_state_stride(capRow, maxCol);
    sp.statesPerProblem = kb::small_states_per_problem(k, nw, maxCol);
    sp.weights = 1;
    sp.condition = condition ? 1 : 0;
    sp.gate = bruteForce ? 0 : 1;
    sp.probs = mProbs ? mProbs : reinterpret_cast<double *>(dout);
    sp.prof = ctx->prof;
    if (B == 1 && costOff[0] == 0 && probOff[0] == 0) {  // the per-frame call: shape in the kernel arguments
        sp.imm = 1;
        sp.immRow = nRow[0];
        sp.immCol = nM[0];
        sp.immL = nL[0];
        sp.costOff = nullptr;
        sp.probOff = nullptr;
    }
    unsigned char *hout = static_cast<unsigned char *>(ctx->pinOut.host);
    volatile int *hdone = reinterpret_cast<volatile int *>(hout + outBytes);
    const int32_t *hnf = reinterpret_cast<const int32_t *>(hout + probBytes);
    const int32_t *htie = reinterpret_cast<const int32_t *>(hout + probBytes + nfBytes);
    bool anyUnfit = false;
    // First the exhaustive kernel where the whole batch qualifies (tiny_takes); a frame it hands back (-2: thousands of equal
    // gains at slot k, or -- without conditioning -- a block that is not a conditioned one) sends the batch through the fused
    // enumeration kernel after all, whose own -2 (more rows kept than it takes) goes to the general pipeline.
    // 1: the exhaustive kernel, 2: the bounded walk, 0: the fused enumeration kernel
    int fast = !allowFast ? 0 : tiny_takes(ctx, B, nL, nM, kT, condition, bruteForce) ? 1 : bnb_takes(ctx, B, nL, nM, kT, bruteForce) ? 2 : 0;
    for (;;) {
        // (a few frames only: with hundreds of workgroups the counter's system-scope atomics cost more than the wake-up saves)
        if (zeroCopy && !ctx->noPoll && B <= 8) {
            *hdone = 0;
            sp.done = reinterpret_cast<int *>(dout + outBytes);
        }
        hipError_t e = fast == 1   ? kb::launch_kbest_tiny(sp, B, B > 2 * ctx->nCU, ctx->stream)
                       : fast == 2 ? kb::launch_kbest_bnb(sp, B, bnb_many(ctx, B), ctx->stream)
                                   : kb::launch_kbest_small(sp, B, nw, ctx->stream);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "association kernel launch", e);
        if (!zeroCopy) HIP_TRY(ctx, hipMemcpyAsync(hout, dout, outBytes, hipMemcpyDeviceToHost, ctx->stream));
        if (sp.done) {
            // The results land in this host memory; every workgroup bumps the counter (system-scope release) when its last
            // byte is written.  Polling it skips the runtime's completion path (interrupt / wake-up), which is a tenth of a
            // one-frame call.  After ~2 ms without completion the ordinary wait takes over (it also reports device errors).
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (*hdone != B) {
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (*hdone != B) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        anyUnfit = false;
        for (int b = 0; b < B; b++) anyUnfit = anyUnfit || hnf[b] == -2;
        break;
    }
    if (fast && anyUnfit) {
        // the frames the fast kernel handed back -- and only those -- through the fused enumeration kernel
        if (!mProbs) memcpy(probs, hout, nProb * 8);
        if (nf) memcpy(nf, hnf, (size_t)B * 4);
        std::vector<int> idx;
        for (int b = 0; b < B; b++)
            if (hnf[b] == -2) idx.push_back(b);
        for (int b = 0; b < B; b++)
            if (hnf[b] < 0 && hnf[b] != -2) return fail(ctx, KBEST_ERR_INTERNAL, "association kernel: a frame came back with nf < 0");
        const int n = (int)idx.size();
        if (tie)
            for (int b = 0; b < B; b++) tie[b] = tieOn ? htie[b] : 0;
        std::vector<int32_t> sL(n), sM(n), sRow(n), sNf(n), sTie(n, 0);
        std::vector<int64_t> sCo(n), sPo(n);
        for (int i = 0; i < n; i++) { sL[i] = nL[idx[i]]; sM[i] = nM[idx[i]]; sRow[i] = nRow[idx[i]]; sCo[i] = costOff[idx[i]]; sPo[i] = probOff[idx[i]]; }
        std::vector<int> sub;
        const int rc = weights_small(ctx, n, sL.data(), sM.data(), sRow.data(), cost, d_cost, sCo.data(), kT, probs, sPo.data(), sNf.data(),
                                     condition, bruteForce, rawMaxRow, maxCol, nCost, nProb, &sub, false, sTie.data());
        if (rc != KBEST_OK && rc != 1) return rc;
        if (nf)
            for (int i = 0; i < n; i++) nf[idx[i]] = sNf[i];
        if (tie)
            for (int i = 0; i < n; i++) tie[idx[i]] = sTie[i];
        if (rc == 1) {
            if (!unfit) return 1;
            for (int j : sub) unfit->push_back(idx[j]);
            return 1;
        }
        return KBEST_OK;
    }
    if (anyUnfit && unfit)
        for (int b = 0; b < B; b++)
            if (hnf[b] == -2) unfit->push_back(b);  // a frame that keeps more rows than the fused kernel takes: general pipeline
    if (anyUnfit && !unfit) return 1;
    if (!mProbs) {
        if (allowFast) memcpy(probs, hout, nProb * 8);
        else  // the re-run of the frames a fast kernel handed back: a sub-batch, whose frames' own ranges only are this launch's to write
            for (int b = 0; b < B; b++)
                memcpy(probs + probOff[b], reinterpret_cast<const double *>(hout) + probOff[b], (size_t)nM[b] * (nL[b] + 1) * 8);
    }
    if (nf) memcpy(nf, hnf, (size_t)B * 4);
    if (tie)
        for (int b = 0; b < B; b++) tie[b] = (tieOn && hnf[b] >= 0) ? htie[b] : 0;
    for (int b = 0; b < B; b++)
        if (hnf[b] < 0 && hnf[b] != -2) return fail(ctx, KBEST_ERR_INTERNAL, "association kernel: a frame came back with nf < 0");
    return anyUnfit ? 1 : KBEST_OK;
}

extern "C" int kbest_assoc_probs_batch_f64_dev(kbest_ctx *ctx, int B, int maxRawRow, int maxCol, const int32_t *d_nL,
                                               const int32_t *d_nM, const int32_t *d_nRow, const double *d_cost,
                                               const int64_t *d_costOff, int k, int condition, double *d_probs,
                                               const int64_t *d_probOff, int32_t *d_nf, void *stream)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || k < 1 || maxCol < 1 || maxRawRow < maxCol || !d_nL || !d_nM || !d_nRow || !d_cost || !d_costOff || !d_probs ||
        !d_probOff || !d_nf)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_assoc_probs_batch_f64_dev: bad argument");
    if (B == 0) return KBEST_OK;
    // exact ties (kbest_ties.h): the enumeration kernel enumerates one solution more than the kT it weighs; the flags go where
    // kbest_set_assoc_tie_flags_dev says (an asynchronous entry cannot complete a tied gain level: KBEST_TIE_BOUNDARY without
    // KBEST_TIE_RESOLVED tells the caller to re-run that frame through the host-pointer entry)
    const int kT = k;
    const int capRow = maxRawRow < kb::SMALL_MAX_DIM ? maxRawRow : kb::SMALL_MAX_DIM;
    int nw = 0;
    // (the caller's k against the limits: at k = 1 024, or where k + 1 no longer fits the LDS, the launch runs without the
    //  solution behind the k-th and the frames' flags carry KBEST_TIE_UNCHECKED)
    const bool extraSol = !ctx->noTie && maxCol <= kb::SMALL_MAX_DIM && small_fits(ctx, B, capRow, maxCol, kT + 1, true, nullptr);
    if (extraSol) k = k + 1;
    if (maxCol > kb::SMALL_MAX_DIM || k > kb::SMALL_MAX_K || maxRawRow > kb::SMALL_MAX_RAW_ROWS ||
        (!condition && maxRawRow > kb::SMALL_MAX_DIM) || !small_fits(ctx, B, capRow, maxCol, k, true, &nw))
        return fail(ctx, KBEST_ERR_UNSUPPORTED, "kbest_assoc_probs_batch_f64_dev: frames beyond the fused association kernel "
                                                "(nM <= 32, k <= 1024; without conditioning nL + nM <= 32)");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    int rc = order_behind_last(ctx, s);
    if (rc != KBEST_OK) return rc;
    const Launched mark{ctx, s};
    rc = ensure_states(ctx, small_states_need(B, capRow, maxCol, k, nw), false);  // asynchronous entry: never allocates
    if (rc != KBEST_OK) return rc;
    kb::SmallParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.cost = d_cost;
    sp.costOff = reinterpret_cast<const long long *>(d_costOff);
    sp.probOff = reinterpret_cast<const long long *>(d_probOff);
    sp.nRow = d_nRow;
    sp.nCol = d_nM;
    sp.nL = d_nL;
    sp.maxRow = capRow;
    sp.maxCol = maxCol;
    sp.bnbRow = maxRawRow < kb::BNB_MAX_ROW ? maxRawRow : kb::BNB_MAX_ROW;
    sp.ldRow = capRow;
    sp.ldCol = maxCol;
    sp.k = k;
    sp.kTab = kT;
    sp.tieFlags = ctx->noTie ? nullptr : ctx->assocTieDev;
    sp.tieBase = (!ctx->noTie && !extraSol) ? KBEST_TIE_UNCHECKED : 0;
    sp.useCutoff = 1;  // assignment.cpp:594
    sp.cutoff = 42.0;
    sp.nf = d_nf;
    sp.states = ctx->states;
    sp.stateStride = kb::small_state_stride(capRow, maxCol);
    sp.statesPerProblem = kb::small_states_per_problem(k, nw, maxCol);
    sp.weights = 1;
    sp.condition = condition ? 1 : 0;
    sp.gate = 1;
    sp.probs = d_probs;
    sp.prof = ctx->prof;
    // frame-sized blocks of up to 16 measurements and 64 rows: the bounded walk (kbest_bnb.hip; a frame it hands back has
    // d_nf = -2 like one beyond the fused enumeration kernel: the host-pointer entry re-runs such frames by itself)
    const int bnbThreads = bnb_many(ctx, B) ? 256 : 1024;
    const bool useBnb = !ctx->noBnb && maxRawRow <= kb::BNB_MAX_ROW && maxCol <= kb::BNB_MAX_COL && kT <= kb::bnb_max_k(bnbThreads) &&
                        kb::bnb_lds_bytes(kT, bnbThreads) <= ctx->ldsLimit;
    hipError_t e = useBnb ? kb::launch_kbest_bnb(sp, B, bnb_many(ctx, B), s) : kb::launch_kbest_small(sp, B, nw, s);
    if (useBnb && e == hipSuccess) {
        // what the walk hands back (d_nf = -2: masses of equal gains) is answered by the fused enumeration kernel in a second
        // launch that looks at nothing else: the entry stays total and asynchronous
        sp.onlyUnfit = 1;
        e = kb::launch_kbest_small(sp, B, nw, s);
    }
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "association kernel launch", e);
    return KBEST_OK;
}

extern "C" int kbest_set_assoc_tie_flags_dev(kbest_ctx *ctx, int32_t *d_flags)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    ctx->assocTieDev = d_flags;
    return KBEST_OK;
}

extern "C" int kbest_reserve_assoc(kbest_ctx *ctx, int B, int maxRawRow, int maxCol, int k)
{
    if (!ctx || B < 0 || maxCol < 1 || maxRawRow < maxCol || k < 1) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_reserve_assoc: bad argument");
    if (B == 0) return KBEST_OK;
    const int capRow = maxRawRow < kb::SMALL_MAX_DIM ? maxRawRow : kb::SMALL_MAX_DIM;
    int nw = 0;
    // (what the enumeration kernel enumerates: one more than k where that fits -- exact ties, kbest_ties.h)
    if (!ctx->noTie && maxCol <= kb::SMALL_MAX_DIM && small_fits(ctx, B, capRow, maxCol, k + 1, true, nullptr)) k = k + 1;
    if (maxCol > kb::SMALL_MAX_DIM || !small_fits(ctx, B, capRow, maxCol, k, true, &nw))
        return fail(ctx, KBEST_ERR_UNSUPPORTED, "kbest_reserve_assoc: frames beyond the fused association kernel");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    return ensure_states(ctx, small_states_need_upto(ctx, B, capRow, maxCol, k, true), true);
}

// tie (optional): [B] KBEST_TIE_* per frame.  tieExtra > 0: the general pipeline enumerates k + tieExtra solutions and weighs the
// first k of them in the canonical order -- how a frame whose k-th and (k+1)-th gains are equal gets the one answer (kbest_ties.h).
// tieExtra < 0: the general pipeline with the reference-order kernel as its enumeration, whatever the context says -- how such a frame
// gets the REFERENCE's answer (kbest_set_reference_order(ctx, 2)); its frames come back flagged KBEST_TIE_REFERENCE.
static int weights_pipeline(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *cost,
                            const int64_t *costOff, int k, double *probs, const int64_t *probOff, int32_t *nf,
                            bool condition, const QuadricHost *quad = nullptr, bool bruteForce = false, bool allowSmall = true,
                            int32_t *tie = nullptr, int tieExtra = 0)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || (k < 1 && !(quad && k == 0)) || !nL || !nM || (!cost && !quad) || !costOff || !probs || !probOff)
        return fail(ctx, KBEST_ERR_BAD_ARG, "weights: bad argument");
    if (B == 0) return KBEST_OK;
    int maxRow = 1, maxCol = 1;
    size_t nCost = 0, nProb = 0;
    std::vector<int32_t> nRow(B);
    for (int b = 0; b < B; b++) {
        // (a frame without measurements is legal and empty: getAssignmentProbs returns nothing for it, assignment.cpp:50-51)
        if (nM[b] < 0 || nL[b] < 0) return fail(ctx, KBEST_ERR_BAD_ARG, "weights: need nM >= 0, nL >= 0");
        nRow[b] = nL[b] + nM[b];
        if (nRow[b] > maxRow) maxRow = nRow[b];
        if (nM[b] > maxCol) maxCol = nM[b];
        const size_t ce = (size_t)costOff[b] + (size_t)nRow[b] * nM[b];
        const size_t pe = (size_t)probOff[b] + (size_t)nM[b] * (nL[b] + 1);
        if (ce > nCost) nCost = ce;
        if (pe > nProb) nProb = pe;
    }
    if (maxCol > KBEST_MAX_DIM_EXACT) return fail(ctx, KBEST_ERR_UNSUPPORTED, "nM > KBEST_MAX_DIM_EXACT");
    // With conditioning the RAW matrix may have any number of rows (all landmarks of the map); only what
    // conditionCosts keeps must fit the solver, and a frame where it does not comes back with nf = -1.
    const int rawMaxRow = maxRow;
    if (!condition && maxRow > KBEST_MAX_DIM_EXACT) return fail(ctx, KBEST_ERR_UNSUPPORTED, "nL + nM > KBEST_MAX_DIM_EXACT");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (kbest_set_reference_order: the k best in the reference's own order of operations -- the general pipeline with the reference-order
    //  kernel as its enumeration; the fused kernels have an order of ties of their own)
    const bool refRerun = tieExtra < 0;
    if (refRerun) tieExtra = 0;
    const bool refOrder = ctx->refOrder == 1 || refRerun;
    if (allowSmall && !refOrder && !quad && k >= 1 && maxCol <= kb::SMALL_MAX_DIM && rawMaxRow <= kb::SMALL_MAX_RAW_ROWS) {
        // the frame-sized case: one fused launch (kbest_small.hip); the frames that keep more rows than it takes -- and
        // only those -- go through the general pipeline below
        std::vector<int> unfit;
        std::vector<int32_t> tfl((size_t)B, 0);
        const int rc = weights_small(ctx, B, nL, nM, nRow.data(), cost, nullptr, costOff, k, probs, probOff, nf, condition,
                                     bruteForce, rawMaxRow, maxCol, nCost, nProb, &unfit, true, tfl.data());
        if (rc != 1 && rc != KBEST_OK) return rc;
        // a sub-batch through the general pipeline: the frames the fused kernels do not take (extra = 0), then the frames whose
        // k-th and (k+1)-th gains are equal (extra = KBEST_TIE_CAP: their gain level is completed)
        auto rerun = [&](const std::vector<int> &ix, int extra) -> int {
            const int Bs = (int)ix.size();
            std::vector<int32_t> sL(Bs), sM(Bs), sNf(Bs), sT(Bs, 0);
            std::vector<int64_t> sCo(Bs), sPo(Bs);
            for (int i = 0; i < Bs; i++) { sL[i] = nL[ix[i]]; sM[i] = nM[ix[i]]; sCo[i] = costOff[ix[i]]; sPo[i] = probOff[ix[i]]; }
            const int rc2 = weights_pipeline(ctx, Bs, sL.data(), sM.data(), cost, sCo.data(), k, probs, sPo.data(), sNf.data(), condition,
                                             nullptr, bruteForce, false, sT.data(), extra);
            if (rc2 != KBEST_OK) return rc2;
            for (int i = 0; i < Bs; i++) {
                if (nf) nf[ix[i]] = sNf[i];
                tfl[ix[i]] = sT[i];
            }
            return KBEST_OK;
        };
        if (rc == 1 && !unfit.empty()) {
            const int rc2 = rerun(unfit, 0);
            if (rc2 != KBEST_OK) return rc2;
        } else if (rc == 1) {
            goto general;  // (the fused kernel takes none of this batch's shapes)
        }
        if (ctx->refOrder == 2) {  // the frames with a tie at slot k: the reference's own k best (and so its weights)
            std::vector<int> tied;
            for (int b = 0; b < B; b++)
                if (tfl[b] & KBEST_TIE_BOUNDARY) tied.push_back(b);
            if (!tied.empty()) {
                const int rc2 = rerun(tied, -1);
                if (rc2 != KBEST_OK) return rc2;
            }
            if (tie) memcpy(tie, tfl.data(), (size_t)B * 4);
            return KBEST_OK;
        }
        // (64, then 256, then KBEST_TIE_CAP solutions beyond k: until the level ends inside the table)
        for (int extra : {64, 256, 1024, KBEST_TIE_CAP}) {
            std::vector<int> tied;
            for (int b = 0; b < B; b++)
                if ((tfl[b] & KBEST_TIE_BOUNDARY) && !(tfl[b] & KBEST_TIE_RESOLVED) && (extra == 64 ? !(tfl[b] & KBEST_TIE_UNRESOLVED) : true)) tied.push_back(b);
            if (tied.empty()) break;
            const int rc2 = rerun(tied, extra);
            if (rc2 != KBEST_OK) {
                if (extra == 64) return rc2;
                break;  // (a larger re-run that fails leaves the answer of the one before: the frames stay KBEST_TIE_UNRESOLVED)
            }
        }
        if (tie) memcpy(tie, tfl.data(), (size_t)B * 4);
        return KBEST_OK;
    }
general:
    // The five per-problem index arrays travel as ONE block (one copy instead of five), and nf sits right behind
    // the probabilities (one copy back instead of two): per-frame calls are dominated by call overheads.
    DevBuf dCost, dCond, dMeta, dGood, dCondL, dRowIdx, dR4C, dGain, dOut;
    const size_t B8 = ((size_t)B * 8 + 15) & ~(size_t)15, B4 = ((size_t)B * 4 + 15) & ~(size_t)15;
    std::vector<unsigned char> meta(2 * B8 + 3 * B4);
    memcpy(meta.data(), costOff, (size_t)B * 8);
    memcpy(meta.data() + B8, probOff, (size_t)B * 8);
    memcpy(meta.data() + 2 * B8, nRow.data(), (size_t)B * 4);
    memcpy(meta.data() + 2 * B8 + B4, nM, (size_t)B * 4);
    memcpy(meta.data() + 2 * B8 + 2 * B4, nL, (size_t)B * 4);
    const size_t probBytes = (nProb * 8 + 15) & ~(size_t)15;
    HIP_TRY(ctx, dCost.alloc(ctx, nCost * 8));
    HIP_TRY(ctx, dMeta.alloc(ctx, meta.size()));
    HIP_TRY(ctx, dOut.alloc(ctx, probBytes + (size_t)B * 4));
    HIP_TRY(ctx, hipMemcpy(dMeta.p, meta.data(), meta.size(), hipMemcpyHostToDevice));
    unsigned char *mb = dMeta.as<unsigned char>();
    Sub dOff{mb}, dPOff{mb + B8}, dNR{mb + 2 * B8}, dNC{mb + 2 * B8 + B4}, dNL{mb + 2 * B8 + 2 * B4};
    Sub dProbs{dOut.p}, dNf{dOut.as<unsigned char>() + probBytes};
    HIP_TRY(ctx, hipMemsetAsync(dProbs.p, 0, nProb * 8, ctx->stream));
    DevBuf dLM, dLC, dMM, dMC, dLOff, dMOff;
    if (!quad) {
        HIP_TRY(ctx, hipMemcpy(dCost.p, cost, nCost * 8, hipMemcpyHostToDevice));
    } else {
        // build the cost blocks on the device (computeQuadricCostMatrix, assignment.cpp:705-722)
        std::vector<long long> lo(B), mo(B);
        long long sl = 0, sm = 0;
        for (int b = 0; b < B; b++) { lo[b] = sl; mo[b] = sm; sl += nL[b]; sm += nM[b]; }
        HIP_TRY(ctx, dLM.alloc(ctx, (size_t)sl * 24));
        HIP_TRY(ctx, dLC.alloc(ctx, (size_t)sl * 72));
        HIP_TRY(ctx, dMM.alloc(ctx, (size_t)sm * 24));
        HIP_TRY(ctx, dMC.alloc(ctx, (size_t)sm * 72));
        HIP_TRY(ctx, dLOff.alloc(ctx, (size_t)B * 8));
        HIP_TRY(ctx, dMOff.alloc(ctx, (size_t)B * 8));
        HIP_TRY(ctx, hipMemcpy(dLM.p, quad->landMean, (size_t)sl * 24, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dLC.p, quad->landCov, (size_t)sl * 72, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dMM.p, quad->measMean, (size_t)sm * 24, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dMC.p, quad->measCov, (size_t)sm * 72, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dLOff.p, lo.data(), (size_t)B * 8, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(dMOff.p, mo.data(), (size_t)B * 8, hipMemcpyHostToDevice));
        kb::QuadricParams q;
        q.nL = dNL.as<int>();
        q.nM = dNC.as<int>();
        q.landOff = dLOff.as<long long>();
        q.measOff = dMOff.as<long long>();
        q.landMean = dLM.as<double>();
        q.landCov = dLC.as<double>();
        q.measMean = dMM.as<double>();
        q.measCov = dMC.as<double>();
        q.gate = quad->gate;
        q.cost = dCost.as<double>();
        q.costOff = dOff.as<long long>();
        hipError_t e = kb::launch_quadric_costs(q, B, ctx->stream);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "quadric cost kernel launch", e);
        if (k == 0) {  // cost construction only
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipMemcpy(probs, dCost.p, nCost * 8, hipMemcpyDeviceToHost));
            return KBEST_OK;
        }
        if (!refOrder && maxCol <= kb::SMALL_MAX_DIM && rawMaxRow <= kb::SMALL_MAX_RAW_ROWS) {
            // frame-sized: the cost blocks the quadric kernel just built go straight into the fused association kernel
            const int rc = weights_small(ctx, B, nL, nM, nRow.data(), nullptr, dCost.as<double>(), costOff, k, probs, probOff, nf,
                                         condition, bruteForce, rawMaxRow, maxCol, nCost, nProb);
            if (rc != 1) return rc;
        }
    }
    const double *solveCost = dCost.as<double>();
    const int32_t *solveRows = dNR.as<int32_t>();
    const int *weightNL = dNL.as<int>();
    if (condition) {
        HIP_TRY(ctx, dCond.alloc(ctx, nCost * 8));
        HIP_TRY(ctx, dGood.alloc(ctx, (size_t)B * 4));
        HIP_TRY(ctx, dCondL.alloc(ctx, (size_t)B * 4));
        HIP_TRY(ctx, dRowIdx.alloc(ctx, (size_t)B * rawMaxRow * 4));
        kb::CondParams c;
        c.cost = dCost.as<double>();
        c.costOff = dOff.as<long long>();
        c.nRow = dNR.as<int>();
        c.nCol = dNC.as<int>();
        c.out = dCond.as<double>();
        c.goodRows = dGood.as<int>();
        c.condL = dCondL.as<int>();
        c.rowIdx = dRowIdx.as<int>();
        c.maxRow = rawMaxRow;
        hipError_t e = kb::launch_condition(c, B, ctx->stream);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "condition kernel launch", e);
        solveCost = dCond.as<double>();
        solveRows = dGood.as<int32_t>();
        weightNL = dCondL.as<int>();
        if (rawMaxRow > KBEST_MAX_DIM) {
            // size the solver for what conditionCosts kept, not for the raw map (the general-size kernel is several
            // times slower per row than the LDS kernels: worth one round trip)
            std::vector<int32_t> good(B);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipMemcpy(good.data(), dGood.p, (size_t)B * 4, hipMemcpyDeviceToHost));
            maxRow = maxCol;
            for (int b = 0; b < B; b++)
                if (good[b] > maxRow) maxRow = good[b];
            if (maxRow > KBEST_MAX_DIM_EXACT) maxRow = KBEST_MAX_DIM_EXACT;  // frames beyond it come back with nf = -1
        }
        // (up to 64 raw rows the LDS kernel takes whatever is kept: the launch is sized from the raw row count and
        //  nothing waits for the conditioning kernel)
    }
    // exact ties (kbest_ties.h): the table comes back in the canonical order with a flag per frame; with tieExtra it holds
    // k + tieExtra solutions of which the first k are weighed (a completed gain level at slot k)
    const int kEnum = k + tieExtra;
    const size_t nR4C = (size_t)B * kEnum * maxCol, nG = (size_t)B * kEnum;
    HIP_TRY(ctx, dR4C.alloc(ctx, nR4C * 4));
    HIP_TRY(ctx, dGain.alloc(ctx, nG * 8));
    DevBuf dTie;
    HIP_TRY(ctx, dTie.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, hipMemsetAsync(dTie.p, 0, (size_t)B * 4, ctx->stream));
    kbest_opts o;
    kbest_default_opts(&o);
    o.use_cutoff = bruteForce ? 0 : 1;  // assignment.cpp:594: kBest2DCutoff(..., cutoff = 42); :880: plain kBest2D
    o.cutoff = 42.0;
    o.tie_flags = dTie.as<int32_t>();
    if (refOrder) o.flags |= KBEST_FLAG_REFERENCE_ORDER;
    // the weights only need row4col: no col4row table
    int rc = batch_dev_impl(ctx, &o, B, maxRow, maxCol, solveRows, dNC.as<int32_t>(), solveCost,
                            dOff.as<int64_t>(), kEnum, dR4C.as<int32_t>(), nullptr, dGain.as<double>(),
                            dNf.as<int32_t>(), nullptr, ctx->stream, true);
    if (rc != KBEST_OK) return rc;
    kb::WeightParams w;
    w.nL = weightNL;
    w.nM = dNC.as<int>();
    w.cost = solveCost;
    w.costOff = dOff.as<long long>();
    w.gain = dGain.as<double>();
    w.row4col = dR4C.as<int>();
    w.nf = dNf.as<int>();
    w.probs = dProbs.as<double>();
    w.probOff = dPOff.as<long long>();
    w.k = kEnum;
    w.kUse = k;
    w.maxCol = maxCol;
    w.rowIdx = condition ? dRowIdx.as<int>() : nullptr;
    w.nLout = dNL.as<int>();
    w.maxRow = rawMaxRow;
    w.gate = bruteForce ? 0 : 1;
    w.solveRows = maxRow;
    {
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        hipError_t e = kb::launch_weights(w, B, ctx->stream);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "weights kernel launch", e);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {   // probabilities and counts come back in one copy
        std::vector<unsigned char> out(probBytes + (size_t)B * 4);
        HIP_TRY(ctx, hipMemcpy(out.data(), dOut.p, out.size(), hipMemcpyDeviceToHost));
        if (allowSmall) {
            memcpy(probs, out.data(), nProb * 8);
        } else {  // a subset of a larger batch (the frames the fused kernel passed on): only their own blocks come back
            for (int b = 0; b < B; b++)
                memcpy(probs + probOff[b], out.data() + (size_t)probOff[b] * 8, (size_t)nM[b] * (nL[b] + 1) * 8);
        }
        const int32_t *hnf = reinterpret_cast<const int32_t *>(out.data() + probBytes);
        if (nf)
            for (int b = 0; b < B; b++) nf[b] = hnf[b] > k ? k : hnf[b];
        // a caller without an nf array (the reference-named shims) must not get all-zero probabilities silently:
        // -3 is an engine failure; -1 (the conditioned block is beyond every kernel) is reported through nf when there is one
        for (int b = 0; b < B; b++)
            if (hnf[b] == -3 || (hnf[b] < 0 && !nf))
                return fail(ctx, hnf[b] == -3 ? KBEST_ERR_INTERNAL : KBEST_ERR_UNSUPPORTED, "association weights: a frame came back with nf < 0");
    }
    if (!ctx->noTie) {
        std::vector<int32_t> tfl((size_t)B), hn((size_t)B);
        HIP_TRY(ctx, hipMemcpy(tfl.data(), dTie.p, (size_t)B * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(hn.data(), dNf.p, (size_t)B * 4, hipMemcpyDeviceToHost));
        if (tieExtra > 0) {
            // this WAS the completing run: the level at slot k is complete when the table goes on beyond it
            std::vector<double> g((size_t)B * kEnum);
            HIP_TRY(ctx, hipMemcpy(g.data(), dGain.p, g.size() * 8, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++) {
                const double *gb = g.data() + (size_t)b * kEnum;
                const bool boundary = hn[b] > k && gb[k] == gb[k - 1];
                const bool complete = (hn[b] < kEnum || gb[kEnum - 1] != gb[k - 1]) && !(tfl[b] & KBEST_TIE_UNORDERED);
                tfl[b] = (tfl[b] & (KBEST_TIE_INSIDE | KBEST_TIE_UNORDERED)) |
                         (boundary ? (KBEST_TIE_BOUNDARY | (complete ? KBEST_TIE_RESOLVED : KBEST_TIE_UNRESOLVED)) : 0);
            }
        } else {
            std::vector<int> tied;
            for (int b = 0; b < B; b++)
                if (tfl[b] & KBEST_TIE_BOUNDARY) tied.push_back(b);
            if (refRerun) {
                for (int b = 0; b < B; b++) tfl[b] = KBEST_TIE_REFERENCE;
            } else if (!tied.empty() && !quad) {
                for (int step : {64, 256, 1024, KBEST_TIE_CAP}) {
                    const int extra = ctx->refOrder == 2 ? -1 : step;  // (kbest_set_reference_order(ctx, 2): ONE re-run, on the reference-order kernel)
                    const int Bs = (int)tied.size();
                    std::vector<int32_t> sL(Bs), sM(Bs), sNf(Bs), sT(Bs, 0);
                    std::vector<int64_t> sCo(Bs), sPo(Bs);
                    for (int i = 0; i < Bs; i++) { sL[i] = nL[tied[i]]; sM[i] = nM[tied[i]]; sCo[i] = costOff[tied[i]]; sPo[i] = probOff[tied[i]]; }
                    const int rc2 = weights_pipeline(ctx, Bs, sL.data(), sM.data(), cost, sCo.data(), k, probs, sPo.data(), sNf.data(), condition,
                                                     nullptr, bruteForce, false, sT.data(), extra);
                    if (rc2 != KBEST_OK) {
                        if (extra == 64 || extra < 0) return rc2;
                        break;
                    }
                    std::vector<int> still;
                    for (int i = 0; i < Bs; i++) {
                        tfl[tied[i]] = sT[i];
                        if ((sT[i] & KBEST_TIE_BOUNDARY) && !(sT[i] & KBEST_TIE_RESOLVED)) still.push_back(tied[i]);
                    }
                    tied.swap(still);
                    if (tied.empty()) break;
                }
            } else {
                for (int b : tied) tfl[b] |= KBEST_TIE_UNRESOLVED;
            }
        }
        if (tie) memcpy(tie, tfl.data(), (size_t)B * 4);
    }
    return KBEST_OK;
}

// the association entries have no kbest_opts to carry a flags pointer: their flags are kept with the context
static int weights_entry(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *cost, const int64_t *costOff, int k,
                         double *probs, const int64_t *probOff, int32_t *nf, bool condition, const QuadricHost *quad, bool bruteForce)
{
    std::vector<int32_t> t((size_t)(B > 0 ? B : 0), 0);
    const int rc = weights_pipeline(ctx, B, nL, nM, cost, costOff, k, probs, probOff, nf, condition, quad, bruteForce, true, t.data());
    if (ctx && rc == KBEST_OK) {
        std::lock_guard<std::mutex> lock(ctx->tieMu);
        ctx->lastTie.swap(t);
    }
    return rc;
}

long long kbest_relay_launches(kbest_ctx *ctx)
{
    if (!ctx) return -1;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    return ctx->relayLaunches;
}

int kbest_set_reference_order(kbest_ctx *ctx, int on)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (on < 0 || on > 2) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_set_reference_order: 0, 1 or 2");
    ctx->refOrder = on;
    return KBEST_OK;
}

int kbest_last_route(kbest_ctx *ctx)
{
    if (!ctx) return -1;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    return ctx->lastRoute;
}

int kbest_last_tie_flags(kbest_ctx *ctx, int32_t *flags, int cap)
{
    if (!ctx || (cap > 0 && !flags)) return KBEST_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(ctx->tieMu);
    const int n = (int)ctx->lastTie.size();
    for (int i = 0; i < n && i < cap; i++) flags[i] = ctx->lastTie[i];
    return n;
}

int kbest_weights_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *cost,
                            const int64_t *costOff, int k, double *probs, const int64_t *probOff, int32_t *nf)
{
    return weights_entry(ctx, B, nL, nM, cost, costOff, k, probs, probOff, nf, false, nullptr, false);
}

int kbest_bruteforce_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *cost,
                                     const int64_t *costOff, int k, double *probs, const int64_t *probOff, int32_t *nf)
{
    return weights_entry(ctx, B, nL, nM, cost, costOff, k, probs, probOff, nf, false, nullptr, true);
}

int kbest_assoc_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *cost,
                                const int64_t *costOff, int k, double *probs, const int64_t *probOff, int32_t *nf)
{
    return weights_entry(ctx, B, nL, nM, cost, costOff, k, probs, probOff, nf, true, nullptr, false);
}

static void packed_cost_offsets(int B, const int32_t *nL, const int32_t *nM, std::vector<int64_t> &off)
{
    off.resize(B);
    int64_t s = 0;
    for (int b = 0; b < B; b++) { off[b] = s; s += (int64_t)(nL[b] + nM[b]) * nM[b]; }
}

int kbest_quadric_costs_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *landMean,
                            const double *landCov, const double *measMean, const double *measCov, double gate,
                            double *cost)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || !nL || !nM || !landMean || !landCov || !measMean || !measCov || !cost)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_quadric_costs_f64: bad argument");
    if (B == 0) return KBEST_OK;
    std::vector<int64_t> off;
    packed_cost_offsets(B, nL, nM, off);
    QuadricHost q{landMean, landCov, measMean, measCov, gate};
    // k = 0: build the cost blocks only; they come back through the `probs` argument
    return weights_pipeline(ctx, B, nL, nM, nullptr, off.data(), 0, cost, off.data(), nullptr, false, &q);
}

int kbest_quadric_assoc_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM,
                                        const double *landMean, const double *landCov, const double *measMean,
                                        const double *measCov, double gate, int k, double *probs,
                                        const int64_t *probOff, int32_t *nf)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || !nL || !nM || !landMean || !landCov || !measMean || !measCov)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_quadric_assoc_probs_batch_f64: bad argument");
    if (B == 0) return KBEST_OK;
    std::vector<int64_t> off;
    packed_cost_offsets(B, nL, nM, off);
    QuadricHost q{landMean, landCov, measMean, measCov, gate};
    return weights_entry(ctx, B, nL, nM, nullptr, off.data(), k, probs, probOff, nf, true, &q, false);
}

int kbest_bb_match_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nR, const double *boxL,
                             const double *boxR, double gate, int32_t *assign)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || !nL || !nR || !boxL || !boxR || !assign)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_bb_match_batch_f64: bad argument");
    if (B == 0) return KBEST_OK;
    std::vector<long long> offL(B), offR(B), costOff(B);
    std::vector<int32_t> nRow(B);
    long long sl = 0, sr = 0, sc = 0;
    int maxRow = 1, maxCol = 1;
    for (int b = 0; b < B; b++) {
        // (asgnBB returns an empty vector for a frame without left boxes and all -1 for one without right boxes,
        //  assignment.cpp:730-732: both are legal frames, not errors)
        if (nL[b] < 0 || nR[b] < 0) return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_bb_match_batch_f64: need nL >= 0, nR >= 0");
        offL[b] = sl; offR[b] = sr; costOff[b] = sc;
        nRow[b] = nR[b] + nL[b];
        sl += nL[b]; sr += nR[b]; sc += (long long)nRow[b] * nL[b];
        if (nRow[b] > maxRow) maxRow = nRow[b];
        if (nL[b] > maxCol) maxCol = nL[b];
    }
    if (maxRow > KBEST_MAX_DIM_WIDE) return fail(ctx, KBEST_ERR_UNSUPPORTED, "nL + nR > KBEST_MAX_DIM_WIDE");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dBL, dBR, dOL, dOR, dOC, dNL, dNRt, dNRow, dCost, dR4C, dC4R, dGain, dNf, dAsg;
    HIP_TRY(ctx, dBL.alloc(ctx, (size_t)sl * 40));
    HIP_TRY(ctx, dBR.alloc(ctx, (size_t)sr * 40));
    HIP_TRY(ctx, dOL.alloc(ctx, (size_t)B * 8));
    HIP_TRY(ctx, dOR.alloc(ctx, (size_t)B * 8));
    HIP_TRY(ctx, dOC.alloc(ctx, (size_t)B * 8));
    HIP_TRY(ctx, dNL.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dNRt.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dNRow.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dCost.alloc(ctx, (size_t)sc * 8));
    HIP_TRY(ctx, dR4C.alloc(ctx, (size_t)B * maxCol * 4));
    HIP_TRY(ctx, dC4R.alloc(ctx, (size_t)B * maxRow * 4));
    HIP_TRY(ctx, dGain.alloc(ctx, (size_t)B * 8));
    HIP_TRY(ctx, dNf.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dAsg.alloc(ctx, (size_t)sl * 4));
    HIP_TRY(ctx, hipMemcpy(dBL.p, boxL, (size_t)sl * 40, hipMemcpyHostToDevice));
    if (sr) HIP_TRY(ctx, hipMemcpy(dBR.p, boxR, (size_t)sr * 40, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dOL.p, offL.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dOR.p, offR.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dOC.p, costOff.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dNL.p, nL, (size_t)B * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dNRt.p, nR, (size_t)B * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dNRow.p, nRow.data(), (size_t)B * 4, hipMemcpyHostToDevice));
    kb::BoxParams bp;
    bp.nL = dNL.as<int>();
    bp.nR = dNRt.as<int>();
    bp.offL = dOL.as<long long>();
    bp.offR = dOR.as<long long>();
    bp.boxL = dBL.as<double>();
    bp.boxR = dBR.as<double>();
    bp.gate = gate;
    bp.cost = dCost.as<double>();
    bp.costOff = dOC.as<long long>();
    bp.assign = dAsg.as<int>();
    hipError_t e = kb::launch_bb_costs(bp, B, ctx->stream);  // computeBBCostMatrix, assignment.cpp:777-797
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "bounding-box cost kernel launch", e);
    kbest_opts o;
    kbest_default_opts(&o);
    o.maximize = 1;  // assignment.cpp:749-750: kBest2D(k = 1, maximize = true)
    int rc = batch_dev_impl(ctx, &o, B, maxRow, maxCol, dNRow.as<int32_t>(), dNL.as<int32_t>(), dCost.as<double>(),
                            dOC.as<int64_t>(), 1, dR4C.as<int32_t>(), dC4R.as<int32_t>(), dGain.as<double>(),
                            dNf.as<int32_t>(), nullptr, ctx->stream, true);
    if (rc != KBEST_OK) return rc;
    e = kb::launch_bb_assign(bp, dR4C.as<int>(), dNf.as<int>(), 1, maxCol, B, ctx->stream);
    if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "bounding-box assign kernel launch", e);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(assign, dAsg.p, (size_t)sl * 4, hipMemcpyDeviceToHost));
    return KBEST_OK;
}

int kbest_condition_costs_f64(kbest_ctx *ctx, int B, const int32_t *nRow, const int32_t *nCol, const double *cost,
                              const int64_t *costOff, double *out, int32_t *goodRows, int32_t *rowIdx, int maxRow)
{
    if (!ctx) return KBEST_ERR_BAD_ARG;
    if (B < 0 || !nRow || !nCol || !cost || !costOff || !out || !goodRows || !rowIdx || maxRow < 1)
        return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_condition_costs_f64: bad argument");
    if (B == 0) return KBEST_OK;
    size_t nCost = 0;
    for (int b = 0; b < B; b++) {
        if (nRow[b] < 1 || nCol[b] < 1 || nRow[b] > maxRow || nCol[b] > KBEST_MAX_DIM_WIDE)
            return fail(ctx, KBEST_ERR_BAD_ARG, "kbest_condition_costs_f64: shape out of range");
        const size_t ce = (size_t)costOff[b] + (size_t)nRow[b] * nCol[b];
        if (ce > nCost) nCost = ce;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dCost, dOut, dOff, dNR, dNC, dGood, dRowIdx;
    HIP_TRY(ctx, dCost.alloc(ctx, nCost * 8));
    HIP_TRY(ctx, dOut.alloc(ctx, nCost * 8));
    HIP_TRY(ctx, dOff.alloc(ctx, (size_t)B * 8));
    HIP_TRY(ctx, dNR.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dNC.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dGood.alloc(ctx, (size_t)B * 4));
    HIP_TRY(ctx, dRowIdx.alloc(ctx, (size_t)B * maxRow * 4));
    HIP_TRY(ctx, hipMemcpy(dCost.p, cost, nCost * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dOff.p, costOff, (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dNR.p, nRow, (size_t)B * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dNC.p, nCol, (size_t)B * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemsetAsync(dOut.p, 0, nCost * 8, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dRowIdx.p, 0xFF, (size_t)B * maxRow * 4, ctx->stream));
    kb::CondParams c;
    c.cost = dCost.as<double>();
    c.costOff = dOff.as<long long>();
    c.nRow = dNR.as<int>();
    c.nCol = dNC.as<int>();
    c.out = dOut.as<double>();
    c.goodRows = dGood.as<int>();
    c.condL = nullptr;
    c.rowIdx = dRowIdx.as<int>();
    c.maxRow = maxRow;
    {
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        hipError_t e = kb::launch_condition(c, B, ctx->stream);
        if (e != hipSuccess) return fail(ctx, KBEST_ERR_HIP, "condition kernel launch", e);
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, dOut.p, nCost * 8, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(goodRows, dGood.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(rowIdx, dRowIdx.p, (size_t)B * maxRow * 4, hipMemcpyDeviceToHost));
    return KBEST_OK;
}

}  // extern "C"
