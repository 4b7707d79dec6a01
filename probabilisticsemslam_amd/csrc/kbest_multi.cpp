// kbest_multi.cpp -- the multi-device entry points of include/kbest_c.h: one engine context and one stream per GPU,
// contiguous block sharding of the batch, and the RCCL all-gather (over xGMI) of the packed per-device result tables
// (gain[k], row4col[k*M], nf per matrix -- SURVEY 8(e)) that leaves every device with the same global k-best table.
// What travels is as narrow as the problem allows: every index of a problem of up to 127 rows fits a byte, so row4col
// crosses the links as int8 (8 + M instead of 8 + 4 M bytes per solution: 14.7 MB instead of 54 MB per device for 1 024 x 64x64,
// k = 200).
// This is the C++ side of BASELINE.json's config 4 ("sharded across 8 MI355X via RCCL top-k allgather"): a host
// program written like the reference (one process, plain C++) shards without a Python launcher.
//
// Batch mode: the matrices are independent, so there is NO data-path collective: each device solves its block with the
// same kernels as the single-device entries, writing straight into its packed slice (gain | row4col | nf) of the global
// table; ONE in-place ncclAllGather of the packed slices (send buffer = own slice of the receive buffer) is the only
// exchange.  Subtree mode (few large matrices): every device enumerates its share of the root's subtrees of EVERY matrix; the
// per-shard top-k COSTS (gain[k] + nf: 8 k + 4 bytes per matrix and shard) are all-gathered, every device merges them into the
// global k-best heap (merge_gains_kernel, kbest_merge.hip) -- which also tells it which of ITS OWN assignments made the cut --,
// and one sum all-reduce of a byte table that holds every winner's row at its merged position (k M bytes per matrix, whatever
// the number of shards) completes the table everywhere: the north star's "allgather of per-rank top-k costs into a global k-best
// heap".  Two candidates with exactly the same gain can only be ordered by their assignments: such a call (integer-like costs)
// -- and problems of more than 127 rows -- exchange the whole lists and merge with merge_topk_kernel, as before.  On any failure
// every device stream that was given work is drained and an open RCCL group is closed before the entry returns.
//
// The host side is parallel: one worker thread per device feeds it (upload, launches: the single-device host path of
// kbest_capi.cpp with its pieces, the tables staged in the device's slice) and reads ITS OWN slice of the results back, so no
// device waits for another one's copies; the host times of every step are kept per device (kbest_multi_timeline).
// Several entries of device_ids may name the SAME GPU ("logical devices": the one-GPU tests and the bench's multi-entry line
// exercise the whole host path with G > 1 that way); RCCL cannot span one GPU twice, so such a context exchanges the
// slices by device-to-device copies instead of ncclAllGather -- same buffers, same result.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library stays loadable -- and every single-device entry
// usable -- on a host without RCCL, and a process that already carries another copy of RCCL (PyTorch bundles one) does
// not get a second one forced into its link map unless it asks for the multi-device entries.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kbest_c.h"
#include "kbest_engine.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err)
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("dlopen(librccl): ") + dlerror(); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl: missing symbols";
            return false;
        }
        return true;
    }
};

struct Dev {
    int id = 0;
    kbest_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    bool issued = false;        // work of the current call has been put on `stream`
    hipEvent_t ev = nullptr;    // local exchange (logical devices): this device's slice is ready / its copies are out
    int rc = 0;                 // result of this device's worker in the current call
    std::string werr;
    double t[KBEST_MULTI_STAMPS] = {0, 0, 0, 0, 0, 0};  // host times of the current call, seconds since its entry
    // device buffers, grown on demand
    double *cost = nullptr;
    int32_t *shape = nullptr;   // nRow | nCol of this device's problems
    int32_t *c4r = nullptr;     // col4row of this device's block (not gathered: SURVEY 8(e) exchanges gain, row4col, nf)
    signed char *r8 = nullptr;  // row4col of this device's block as bytes (narrow staging of the host path: what crosses PCIe) when the slice holds int32
    size_t r8B = 0;
    int32_t *r32 = nullptr;     // row4col of this device's block as int32 (what the caller's tables hold) when the slice holds bytes
    size_t r32B = 0;
    unsigned char *packed = nullptr;  // the global table: one packed slice (gain | row4col | nf) per device, identical everywhere after the gather
    double *mGain = nullptr;    // subtree mode: the merged global k best (identical on every device)
    int32_t *mR4C = nullptr, *mNf = nullptr;
    unsigned char *heads = nullptr;   // subtree mode: every device's (gain | nf) of its shards, gathered
    signed char *mRow8 = nullptr;     // ... the winners' rows at their merged positions (own ones, then all: the sum all-reduce)
    int *tied = nullptr;              // ... one word: two candidates of some matrix have exactly the same gain
    int32_t *tieFl = nullptr;         // batch mode: KBEST_TIE_* of this device's problems
    size_t tieFlB = 0;
    size_t costB = 0, shapeB = 0, c4rB = 0, packedB = 0, mGainB = 0, mR4CB = 0, mNfB = 0, headsB = 0, mRow8B = 0, tiedB = 0;
};

}  // namespace

// The per-device host workers: G - 1 threads that live as long as the context (a thread's first HIP call on a device costs
// about a millisecond -- more than a whole share of config 4 --, so they are not created per call), each bound to its device
// once; device 0's job runs on the calling thread.
struct Workers {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cvGo, cvDone;
    const std::function<void(int)> *job = nullptr;
    unsigned long long gen = 0;  // bumped per dispatch
    int pending = 0;
    bool quit = false;
    void start(int G, const std::vector<int> &ids)
    {
        for (int g = 1; g < G; g++)
            th.emplace_back([this, g, id = ids[g]]() {
                (void)hipSetDevice(id);
                unsigned long long seen = 0;
                for (;;) {
                    const std::function<void(int)> *j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cvGo.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        j = job;
                    }
                    (*j)(g);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cvDone.notify_all();
                    }
                }
            });
    }
    void run(const std::function<void(int)> &j)  // job(g) for every device, job(0) on this thread; returns when all are done
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &j;
            pending = (int)th.size();
            gen++;
        }
        cvGo.notify_all();
        j(0);
        std::unique_lock<std::mutex> lk(mu);
        cvDone.wait(lk, [&] { return pending == 0; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cvGo.notify_all();
        for (auto &t : th) t.join();
        th.clear();
    }
};

struct kbest_multi {
    std::vector<Dev> dev;
    Workers workers;
    Rccl rccl;
    std::string err;
    bool local = false;  // device_ids names a GPU more than once: no RCCL communicator, the slices travel by device-to-device copies
    double t0 = 0;       // entry time of the current call
    // shape of the last call (for kbest_multi_tables_agree)
    int lastB = 0, lastK = 0, lastCol = 0, lastMode = 0;
    size_t lastBytes = 0;  // bytes of the packed global table
    size_t lastSent = 0;   // bytes that arrived at one device in the exchanges of the last call (kbest_multi_exchange_bytes)
    std::vector<int32_t> lastTie;  // batch mode: KBEST_TIE_* per problem of the last call (kbest_multi_last_tie_flags)
    int lastPath = 0;      // subtree mode: 1 = gains first (all-gather of the costs + sum all-reduce of the winners' rows), 2 = whole lists
};

namespace {

int mfail(kbest_multi *m, int code, const std::string &what)
{
    static std::mutex mu;  // (the per-device workers may fail at the same time)
    std::lock_guard<std::mutex> lock(mu);
    if (m) m->err = what;
    return code;
}

size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

// One packed slice: gain[n][k] fp64 | row4col[n][k][maxCol] (esz bytes per entry: int8 where every index fits a byte, else
// int32) | nf[n] i32 (each part 16-byte aligned)
struct Slice {
    size_t offGain, offR4C, offNf, bytes;
    Slice(size_t n, int k, int maxCol, size_t esz)
    {
        offGain = 0;
        offR4C = up16(n * (size_t)k * 8);
        offNf = offR4C + up16(n * (size_t)k * maxCol * esz);
        bytes = offNf + up16(n * 4);
    }
};

// Subtree mode, one device's block: gain[spd][B][k] fp64 | nf[spd][B] i32 -- the "head": what the gains-first exchange gathers --
// | row4col[spd][B][k][maxCol] (esz bytes per entry)
struct SubSlice {
    size_t offGain, offNf, offR4C, head, bytes;
    SubSlice(int spd, size_t B, int k, int maxCol, size_t esz)
    {
        offGain = 0;
        offNf = up16((size_t)spd * B * k * 8);
        offR4C = offNf + up16((size_t)spd * B * 4);
        head = offR4C;
        bytes = offR4C + up16((size_t)spd * B * k * maxCol * esz);
    }
};

// Every device stream that has been given work in this call is drained: nothing of a failed call may still read the
// caller's buffers or write device memory when the entry returns.
void drain(kbest_multi *m)
{
    for (auto &d : m->dev)
        if (d.issued) {
            (void)hipSetDevice(d.id);
            (void)hipStreamSynchronize(d.stream);
            d.issued = false;
        }
}

// (runs on the device's own worker thread: a failure is written to the DEVICE's message, never to the shared one)
template <class T> int grow(kbest_multi *, Dev &d, T *&p, size_t &have, size_t need)
{
    if (need <= have) return KBEST_OK;
    if (p) {  // an earlier (asynchronous) call may still use the old buffer
        (void)hipStreamSynchronize(d.stream);
        (void)hipFree(p);
    }
    p = nullptr;
    have = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), need);
    if (e != hipSuccess) {
        d.werr = std::string("hipMalloc: ") + hipGetErrorString(e);
        return KBEST_ERR_NOMEM;
    }
    have = need;
    return KBEST_OK;
}

#define M_TRY(m, expr)                                  \
    do {                                                \
        const int rc_ = (expr);                         \
        if (rc_ != KBEST_OK) { drain(m); return rc_; }  \
    } while (0)
#define M_HIP(m, call)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (call);                                                                                     \
        if (e_ != hipSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
    } while (0)

// ONE all-gather of `bytes` per device, stream-ordered behind each device's kernels: device g contributes send(g) and ends up with
// every device's contribution, h's at recv(g) + h * bytes.  In place when send(g) == recv(g) + g * bytes (the packed slices of the
// global table); the subtree mode's heads go from the device's block into a buffer of their own.  The RCCL group is always closed,
// whatever fails inside it.
template <class Send, class Recv> int gather_bytes(kbest_multi *m, size_t bytes, Send send, Recv recv)
{
    const int G = (int)m->dev.size();
    m->lastSent += (size_t)(G - 1) * bytes;  // what arrives at one device
    if (m->local) {
        // logical devices on one GPU: every device's contribution is copied into every other device's buffer, stream-ordered behind
        // the producer's kernels (event); what orders the copies in front of the NEXT call's writes is the end of this call --
        // every device's stream is synchronised before the entry returns
        for (int g = 0; g < G; g++) {
            Dev &d = m->dev[g];
            if (hipSetDevice(d.id) != hipSuccess || hipEventRecord(d.ev, d.stream) != hipSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, "local gather: event"); }
            d.issued = true;
        }
        for (int g = 0; g < G; g++) {
            Dev &d = m->dev[g];
            for (int h = 0; h < G; h++) {
                unsigned char *dst = recv(g) + (size_t)h * bytes;
                const unsigned char *src = send(h);
                if (dst == src) continue;  // (in place: the device's own slice is where it belongs)
                // device g pulls contribution h once it is ready
                if ((h != g && hipStreamWaitEvent(d.stream, m->dev[h].ev, 0) != hipSuccess) ||
                    hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, d.stream) != hipSuccess) {
                    drain(m);
                    return mfail(m, KBEST_ERR_HIP, "local gather: copy");
                }
            }
        }
        return KBEST_OK;
    }
    ncclResult_t first = ncclSuccess;
    ncclResult_t r = m->rccl.GroupStart();
    if (r != ncclSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, std::string("ncclGroupStart: ") + m->rccl.GetErrorString(r)); }
    for (int g = 0; g < G && first == ncclSuccess; g++) {
        Dev &d = m->dev[g];
        first = m->rccl.AllGather(send(g), recv(g), bytes, ncclChar, d.comm, d.stream);
        d.issued = true;
    }
    r = m->rccl.GroupEnd();
    if (first == ncclSuccess) first = r;
    if (first != ncclSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, std::string("ncclAllGather: ") + m->rccl.GetErrorString(first)); }
    return KBEST_OK;
}

// the packed slices of the global table, in place (batch mode; the whole lists of the subtree mode)
int gather_packed(kbest_multi *m, size_t perDev)
{
    for (auto &d : m->dev) d.t[4] = kb::now_s() - m->t0;
    return gather_bytes(m, perDev, [&](int g) { return m->dev[g].packed + (size_t)g * perDev; }, [&](int g) { return m->dev[g].packed; });
}

// ONE sum all-reduce of every device's byte table mRow8 (n bytes, in place): every entry is non-zero on at most one device -- the
// one whose shard holds the winner of that slot --, so the sum IS the merged table.  Logical devices: device 0 adds the others'
// tables to its own, then every device copies the result.
int allreduce_rows(kbest_multi *m, size_t n)
{
    const int G = (int)m->dev.size();
    m->lastSent += 2 * n * (size_t)(G - 1) / (size_t)G;  // what a ring all-reduce moves into one device: reduce-scatter + all-gather
    if (m->local) {
        for (int g = 0; g < G; g++) {
            Dev &d = m->dev[g];
            if (hipSetDevice(d.id) != hipSuccess || hipEventRecord(d.ev, d.stream) != hipSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, "local all-reduce: event"); }
            d.issued = true;
        }
        Dev &d0 = m->dev[0];
        for (int h = 1; h < G; h++)
            if (hipStreamWaitEvent(d0.stream, m->dev[h].ev, 0) != hipSuccess ||
                kb::launch_add_i8(d0.mRow8, m->dev[h].mRow8, (long long)n, d0.stream) != hipSuccess) {
                drain(m);
                return mfail(m, KBEST_ERR_HIP, "local all-reduce: add");
            }
        if (hipEventRecord(d0.ev, d0.stream) != hipSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, "local all-reduce: event"); }
        for (int g = 1; g < G; g++) {
            Dev &d = m->dev[g];
            if (hipStreamWaitEvent(d.stream, d0.ev, 0) != hipSuccess ||
                hipMemcpyAsync(d.mRow8, d0.mRow8, n, hipMemcpyDeviceToDevice, d.stream) != hipSuccess) {
                drain(m);
                return mfail(m, KBEST_ERR_HIP, "local all-reduce: copy");
            }
        }
        return KBEST_OK;
    }
    ncclResult_t first = ncclSuccess;
    ncclResult_t r = m->rccl.GroupStart();
    if (r != ncclSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, std::string("ncclGroupStart: ") + m->rccl.GetErrorString(r)); }
    for (int g = 0; g < G && first == ncclSuccess; g++) {
        Dev &d = m->dev[g];
        first = m->rccl.AllReduce(d.mRow8, d.mRow8, n, ncclChar, ncclSum, d.comm, d.stream);
        d.issued = true;
    }
    r = m->rccl.GroupEnd();
    if (first == ncclSuccess) first = r;
    if (first != ncclSuccess) { drain(m); return mfail(m, KBEST_ERR_HIP, std::string("ncclAllReduce: ") + m->rccl.GetErrorString(first)); }
    return KBEST_OK;
}

}  // namespace

extern "C" {

int kbest_create_multi(kbest_multi **out, const int *device_ids, int nDev)
{
    if (!out) return KBEST_ERR_BAD_ARG;
    *out = nullptr;
    if (!device_ids || nDev < 1) return KBEST_ERR_BAD_ARG;
    kbest_multi *m = new kbest_multi;
    m->dev.resize(nDev);
    for (int g = 0; g < nDev; g++) {
        m->dev[g].id = device_ids[g];
        int rc = kbest_create(&m->dev[g].ctx, device_ids[g]);
        if (rc != KBEST_OK) { kbest_destroy_multi(m); return rc; }
        if (hipSetDevice(device_ids[g]) != hipSuccess ||
            hipStreamCreateWithFlags(&m->dev[g].stream, hipStreamNonBlocking) != hipSuccess) {
            kbest_destroy_multi(m);
            return KBEST_ERR_NO_DEVICE;
        }
    }
    for (int g = 0; g < nDev; g++)
        for (int h = 0; h < g; h++)
            if (device_ids[g] == device_ids[h]) m->local = true;
    if (m->local) {
        for (int g = 0; g < nDev; g++)
            if (hipSetDevice(device_ids[g]) != hipSuccess || hipEventCreateWithFlags(&m->dev[g].ev, hipEventDisableTiming) != hipSuccess) {
                kbest_destroy_multi(m);
                return KBEST_ERR_HIP;
            }
    } else {
        std::string err;
        if (!m->rccl.load(err)) { kbest_destroy_multi(m); return KBEST_ERR_NO_DEVICE; }
        std::vector<ncclComm_t> comms(nDev);
        if (m->rccl.CommInitAll(comms.data(), nDev, device_ids) != ncclSuccess) { kbest_destroy_multi(m); return KBEST_ERR_HIP; }
        for (int g = 0; g < nDev; g++) m->dev[g].comm = comms[g];
    }
    m->workers.start(nDev, std::vector<int>(device_ids, device_ids + nDev));
    *out = m;
    return KBEST_OK;
}

int kbest_destroy_multi(kbest_multi *m)
{
    if (!m) return KBEST_OK;
    m->workers.stop();
    for (auto &d : m->dev) {
        (void)hipSetDevice(d.id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(d.comm);
        for (void *p : {(void *)d.cost, (void *)d.shape, (void *)d.c4r, (void *)d.packed, (void *)d.mGain, (void *)d.mR4C, (void *)d.mNf, (void *)d.r8, (void *)d.r32, (void *)d.heads,
                        (void *)d.mRow8, (void *)d.tied, (void *)d.tieFl})
            if (p) (void)hipFree(p);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.ev) (void)hipEventDestroy(d.ev);
        if (d.ctx) kbest_destroy(d.ctx);
    }
    delete m;
    return KBEST_OK;
}

int kbest_multi_size(const kbest_multi *m) { return m ? (int)m->dev.size() : 0; }

const char *kbest_multi_last_error(const kbest_multi *m) { return m ? m->err.c_str() : "null context"; }

int kbest_batch_f64_multi_ex(kbest_multi *m, const kbest_opts *opts, int mode, int nShard, int B, int maxRow, int maxCol,
                             const int32_t *nRow, const int32_t *nCol, const double *cost, int k, int32_t *row4col,
                             int32_t *col4row, double *gain, int32_t *nf)
{
    if (!m) return KBEST_ERR_BAD_ARG;
    if (!opts || B < 0 || k < 1 || maxCol < 1 || maxRow < maxCol || !cost || !row4col || !gain || !nf ||
        (nRow == nullptr) != (nCol == nullptr) || (mode != KBEST_MULTI_BATCH && mode != KBEST_MULTI_SUBTREE) || nShard < 0)
        return mfail(m, KBEST_ERR_BAD_ARG, "kbest_batch_f64_multi: bad argument");
    if (opts->flags & KBEST_FLAG_TABLES_I8)  // (the EXCHANGE is in bytes by itself wherever the indices fit them)
        return mfail(m, KBEST_ERR_UNSUPPORTED, "kbest_batch_f64_multi: the caller's tables are int32 (KBEST_FLAG_TABLES_I8 is for the single-device entries)");
    if (nRow)  // the same validation as kbest_batch_f64: a bad shape is an argument error, not a kernel's nf = -1
        for (int b = 0; b < B; b++)
            if (nCol[b] < 1 || nRow[b] < nCol[b] || nRow[b] > maxRow || nCol[b] > maxCol)
                return mfail(m, KBEST_ERR_BAD_ARG, "kbest_batch_f64_multi: shape out of range (need 1 <= numCol <= numRow <= maxRow)");
    if (B == 0) return KBEST_OK;
    const int G = (int)m->dev.size();
    const size_t per = (size_t)maxRow * maxCol;
    // what travels between the devices: row4col as bytes wherever every index fits one (KBEST_MULTI_WIDE=1: int32 as in round 5, A/B)
    const bool forceWideSlices = getenv("KBEST_MULTI_WIDE") != nullptr;
    const size_t esz = (maxRow <= 127 && !forceWideSlices) ? 1 : 4;
    m->t0 = kb::now_s();
    m->lastSent = 0;
    m->lastPath = 0;
    m->lastTie.clear();
    for (auto &d : m->dev) {
        d.issued = false;
        d.rc = KBEST_OK;
        d.werr.clear();
        for (double &x : d.t) x = 0.0;
    }
    // one worker thread per device for the steps that block the host (uploads from the caller's pageable memory, copies back):
    // no device waits for another one's copies.  A worker reports through Dev::rc / werr.
    auto run_workers = [&](const std::function<void(int)> &job) -> int {
        m->workers.run(job);
        for (int g = 0; g < G; g++)
            if (m->dev[g].rc != KBEST_OK) {
                drain(m);
                return mfail(m, m->dev[g].rc, std::string("device ") + std::to_string(m->dev[g].id) + ": " + m->dev[g].werr);
            }
        return KBEST_OK;
    };
    auto wfail = [](Dev &d, int code, const std::string &what) { d.rc = code; d.werr = what; };
#define W_HIP(d, call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) { wfail(d, KBEST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); return; } \
    } while (0)
#define W_TRY(d, expr)                                              \
    do {                                                            \
        const int rc_ = (expr);                                     \
        if (rc_ != KBEST_OK) { d.rc = rc_; return; }  /* (grow() has left its message in d.werr) */ \
    } while (0)

    if (mode == KBEST_MULTI_BATCH) {
        const int pad = (B + G - 1) / G;  // matrices per device (the last devices may hold fewer): equal all-gather counts
        const Slice sl((size_t)pad, k, maxCol, esz);
        m->lastB = B; m->lastK = k; m->lastCol = maxCol; m->lastMode = mode; m->lastBytes = (size_t)G * sl.bytes;
        m->lastTie.assign((size_t)B, 0);
        // 1. every device, in its own thread: its block of cost matrices in, its slice of the global table solved in place and
        //    copied back into the caller's tables -- the single-device host path (kbest_batch_f64: pieces whose uploads,
        //    kernels and copies back overlap), with the tables staged in the device's slice, where they stay for the exchange
        M_TRY(m, run_workers([&](int g) {
            Dev &d = m->dev[g];
            const int b0 = g * pad, nb = (b0 >= B) ? 0 : ((B - b0 < pad) ? B - b0 : pad);
            d.t[0] = kb::now_s() - m->t0;
            W_HIP(d, hipSetDevice(d.id));
            W_TRY(d, grow(m, d, d.packed, d.packedB, (size_t)G * sl.bytes));
            if (col4row) W_TRY(d, grow(m, d, d.c4r, d.c4rB, (size_t)pad * k * maxRow * 4));
            // the block's row4col exists twice on the device: as bytes (what crosses PCIe on the narrow-staged path, and -- esz 1 --
            // what the slice holds) and as int32 (what the caller's tables hold, and -- esz 4 -- the slice)
            if (esz == 1) W_TRY(d, grow(m, d, d.r32, d.r32B, (size_t)pad * k * maxCol * 4));
            else W_TRY(d, grow(m, d, d.r8, d.r8B, (size_t)pad * k * maxCol));
            unsigned char *mine = d.packed + (size_t)g * sl.bytes;
            // slots the kernels do not write (padding problems of the last devices) get defined values: gain 0, row4col -1, nf 0
            d.issued = true;
            if (nb < pad) {
                W_HIP(d, hipMemsetAsync(mine, 0, sl.bytes, d.stream));
                W_HIP(d, hipMemsetAsync(mine + sl.offR4C, 0xFF, (size_t)pad * k * maxCol * esz, d.stream));
                W_HIP(d, hipStreamSynchronize(d.stream));
            }
            if (nb == 0) { d.t[1] = d.t[2] = d.t[3] = kb::now_s() - m->t0; return; }
            W_TRY(d, grow(m, d, d.tieFl, d.tieFlB, (size_t)pad * 4));
            W_HIP(d, hipMemsetAsync(d.tieFl, 0, (size_t)pad * 4, d.stream));
            W_HIP(d, hipStreamSynchronize(d.stream));  // (the pieces of the host path run on streams of the context's own)
            kbest_opts ot = *opts;
            ot.tie_flags = d.tieFl;
            double stamps[2] = {0.0, 0.0};
            const kb::KeepTables keep{esz == 1 ? d.r32 : reinterpret_cast<int32_t *>(mine + sl.offR4C), col4row ? d.c4r : nullptr,
                                      reinterpret_cast<double *>(mine + sl.offGain), reinterpret_cast<int32_t *>(mine + sl.offNf), stamps,
                                      esz == 1 ? reinterpret_cast<signed char *>(mine + sl.offR4C) : d.r8, esz == 1 ? 1 : 0};
            const int rc = kbest_batch_f64_keep(d.ctx, &ot, nb, maxRow, maxCol, nRow ? nRow + b0 : nullptr, nCol ? nCol + b0 : nullptr,
                                                cost + (size_t)b0 * per, nullptr, k, row4col + (size_t)b0 * k * maxCol,
                                                col4row ? col4row + (size_t)b0 * k * maxRow : nullptr, gain + (size_t)b0 * k, nf + b0,
                                                nullptr, &keep);
            d.t[1] = stamps[0] - m->t0;
            d.t[2] = stamps[1] - m->t0;
            if (rc != KBEST_OK) { d.t[3] = kb::now_s() - m->t0; wfail(d, rc, kbest_last_error(d.ctx)); return; }
            // exact ties (kbest_c.h): a gain level that straddles slot k is completed as the synchronous single-device entry does -- in
            // the caller's tables, and in this device's slice of the global table before it travels
            int32_t *fl = m->lastTie.data() + b0;
            W_HIP(d, hipMemcpy(fl, d.tieFl, (size_t)nb * 4, hipMemcpyDeviceToHost));
            bool any = false;
            const int tiedMask = (opts->flags & KBEST_FLAG_CANONICAL_TIES) ? KBEST_TIE_BOUNDARY
                                                                            : (KBEST_TIE_INSIDE | KBEST_TIE_BOUNDARY | KBEST_TIE_UNCHECKED | KBEST_TIE_UNORDERED);
            for (int i = 0; i < nb; i++) any = any || (fl[i] & tiedMask);
            if (any && !(opts->flags & KBEST_FLAG_NO_TIE_RESOLVE)) {
                std::vector<int> changed;
                kb_complete_tie_levels(d.ctx, opts, nb, maxRow, maxCol, nRow ? nRow + b0 : nullptr, nCol ? nCol + b0 : nullptr, cost + (size_t)b0 * per,
                                       nullptr, k, row4col + (size_t)b0 * k * maxCol, col4row ? col4row + (size_t)b0 * k * maxRow : nullptr,
                                       gain + (size_t)b0 * k, fl, &changed);
                std::vector<signed char> r8((size_t)k * maxCol);
                for (int i : changed) {
                    const int32_t *src = row4col + ((size_t)b0 + i) * k * maxCol;
                    unsigned char *dst = mine + sl.offR4C + (size_t)i * k * maxCol * esz;
                    if (esz == 1) {
                        for (size_t j = 0; j < r8.size(); j++) r8[j] = (signed char)src[j];
                        W_HIP(d, hipMemcpy(dst, r8.data(), r8.size(), hipMemcpyHostToDevice));
                    } else {
                        W_HIP(d, hipMemcpy(dst, src, (size_t)k * maxCol * 4, hipMemcpyHostToDevice));
                    }
                    W_HIP(d, hipMemcpy(mine + sl.offGain + (size_t)i * k * 8, gain + ((size_t)b0 + i) * k, (size_t)k * 8, hipMemcpyHostToDevice));
                }
            }
            for (int i = 0; i < nb; i++)
                if ((fl[i] & KBEST_TIE_BOUNDARY) && !(fl[i] & KBEST_TIE_RESOLVED)) fl[i] |= KBEST_TIE_UNRESOLVED;
            d.t[3] = kb::now_s() - m->t0;
        }));
        // 2. the one exchange: ONE all-gather of the packed (gain[k], row4col[k*M], nf) slices; every device then
        //    holds the global table (the host has its results already: each device's own slice came back in step 1)
        M_TRY(m, gather_packed(m, sl.bytes));
        drain(m);
        for (auto &d : m->dev) d.t[5] = kb::now_s() - m->t0;
    } else {
        // Subtree mode (few large matrices; SURVEY 8(e), north star): every device holds ALL B matrices; shard s of S expands
        // only the root's children on columns c % S == s and enumerates its own k best (slot 0: the root).  Shards are dealt
        // to the devices round robin: with S > G a device runs several shards one after the other (and one device can
        // stand in for several: the one-GPU test of this path).  Then the exchange, gains first (the comment at the top).
        const int S = nShard > 0 ? nShard : G;
        const int spd = (S + G - 1) / G;  // shard slots per device
        const SubSlice sl(spd, (size_t)B, k, maxCol, esz);
        const size_t perDev = sl.bytes;
        const size_t nRows8 = (size_t)B * k * maxCol;
        const bool noGainsFirst = getenv("KBEST_MULTI_WHOLE_LISTS") != nullptr;  // (A/B, tests: always the whole lists)
        const bool gainsFirst = esz == 1 && !noGainsFirst;
        m->lastB = B; m->lastK = k; m->lastCol = maxCol; m->lastMode = mode;
        m->lastBytes = 0;
        if (opts->root_col_stride > 1) return mfail(m, KBEST_ERR_BAD_ARG, "kbest_batch_f64_multi: subtree mode sets root_col_offset / stride itself");
        M_TRY(m, run_workers([&](int g) {
            Dev &d = m->dev[g];
            d.t[0] = kb::now_s() - m->t0;
            W_HIP(d, hipSetDevice(d.id));
            W_TRY(d, grow(m, d, d.packed, d.packedB, (size_t)G * perDev));
            W_TRY(d, grow(m, d, d.cost, d.costB, (size_t)B * per * 8));
            W_TRY(d, grow(m, d, d.mGain, d.mGainB, (size_t)B * k * 8));
            W_TRY(d, grow(m, d, d.mR4C, d.mR4CB, (size_t)B * k * maxCol * 4));
            W_TRY(d, grow(m, d, d.mNf, d.mNfB, (size_t)B * 4));
            if (gainsFirst) {
                W_TRY(d, grow(m, d, d.heads, d.headsB, (size_t)G * sl.head));
                W_TRY(d, grow(m, d, d.mRow8, d.mRow8B, nRows8));
                W_TRY(d, grow(m, d, d.tied, d.tiedB, (size_t)16));
            }
            if (nRow) W_TRY(d, grow(m, d, d.shape, d.shapeB, (size_t)2 * B * 4));
            unsigned char *mine = d.packed + (size_t)g * perDev;
            d.issued = true;
            W_HIP(d, hipMemsetAsync(mine, 0, perDev, d.stream));  // (an unused shard slot: nf = 0, no candidates)
            W_HIP(d, hipMemsetAsync(mine + sl.offR4C, 0xFF, (size_t)spd * B * k * maxCol * esz, d.stream));
            W_HIP(d, hipMemsetAsync(d.mR4C, 0xFF, (size_t)B * k * maxCol * 4, d.stream));
            W_HIP(d, hipMemsetAsync(d.mGain, 0, (size_t)B * k * 8, d.stream));
            if (gainsFirst) {
                W_HIP(d, hipMemsetAsync(d.mRow8, 0, nRows8, d.stream));
                W_HIP(d, hipMemsetAsync(d.tied, 0, 16, d.stream));
            }
            d.t[1] = kb::now_s() - m->t0;
            // (from the caller's pageable memory the copy blocks THIS thread until the data is staged; the other devices' threads run)
            W_HIP(d, hipMemcpyAsync(d.cost, cost, (size_t)B * per * 8, hipMemcpyHostToDevice, d.stream));
            if (nRow) {
                W_HIP(d, hipMemcpyAsync(d.shape, nRow, (size_t)B * 4, hipMemcpyHostToDevice, d.stream));
                W_HIP(d, hipMemcpyAsync(d.shape + B, nCol, (size_t)B * 4, hipMemcpyHostToDevice, d.stream));
            }
            int rc = kbest_reserve(d.ctx, B, maxRow, k);
            for (int j = 0; j < spd && rc == KBEST_OK; j++) {
                const int sh = j * G + g;  // round robin
                if (sh >= S) break;
                kbest_opts o = *opts;
                o.root_col_offset = sh;
                o.root_col_stride = S;
                if (esz == 1) o.flags |= KBEST_FLAG_TABLES_I8;
                rc = kbest_batch_f64_dev(d.ctx, &o, B, maxRow, maxCol, nRow ? d.shape : nullptr, nRow ? d.shape + B : nullptr, d.cost,
                                         nullptr, k, reinterpret_cast<int32_t *>(mine + sl.offR4C + (size_t)j * B * k * maxCol * esz), nullptr,
                                         reinterpret_cast<double *>(mine + sl.offGain) + (size_t)j * B * k,
                                         reinterpret_cast<int32_t *>(mine + sl.offNf) + (size_t)j * B, nullptr, d.stream);
                if (j == 0) d.t[2] = kb::now_s() - m->t0;
            }
            d.t[3] = kb::now_s() - m->t0;
            if (rc != KBEST_OK) wfail(d, rc, kbest_last_error(d.ctx));
        }));
        // In the gathered buffers shard s = j * G + g sits in block g at table j: the merges take "shard i" at block i / spd, table
        // i % spd, which enumerates the slots device by device -- any order of the shards gives the same table (ties are ordered by
        // the assignment), but the ROOT is taken from the first slot, which is shard 0 (device 0, j = 0).
        bool whole = !gainsFirst;
        if (gainsFirst) {
            // 1. the top-k COSTS of every shard to every device: ONE all-gather of the heads
            for (auto &d : m->dev) d.t[4] = kb::now_s() - m->t0;
            M_TRY(m, gather_bytes(m, sl.head, [&](int g) { return m->dev[g].packed + (size_t)g * perDev; }, [&](int g) { return m->dev[g].heads; }));
            // 2. the global k-best heap on every device; its own winners' rows into the byte table
            M_TRY(m, run_workers([&](int g) {
                Dev &d = m->dev[g];
                W_HIP(d, hipSetDevice(d.id));
                kb::MergeGainsParams q;
                memset(&q, 0, sizeof(q));
                q.gain = d.heads + sl.offGain;
                q.nf = d.heads + sl.offNf;
                q.blockStride = (long long)sl.head;
                q.spd = spd;
                q.ownRow8 = reinterpret_cast<const signed char *>(d.packed + (size_t)g * perDev + sl.offR4C);
                q.ownLo = g * spd;
                q.ownHi = (g + 1) * spd;
                q.nShard = G * spd;
                q.B = B;
                q.k = k;
                q.maxCol = maxCol;
                q.maximize = opts->maximize;
                q.outGain = d.mGain;
                q.outRow8 = d.mRow8;
                q.outNf = d.mNf;
                q.tied = d.tied;
                const hipError_t e = kb::launch_merge_gains(q, d.stream);
                if (e != hipSuccess) wfail(d, KBEST_ERR_HIP, std::string("merge kernel launch: ") + hipGetErrorString(e));
            }));
            // 3. ONE sum all-reduce completes the rows everywhere
            M_TRY(m, allreduce_rows(m, nRows8));
            // 4. exactly equal gains somewhere?  (the same answer on every device: they merged the same gains)
            int tiedAny = 0;
            for (auto &d : m->dev) {
                int t = 0;
                M_HIP(m, hipSetDevice(d.id));
                M_HIP(m, hipStreamSynchronize(d.stream));
                M_HIP(m, hipMemcpy(&t, d.tied, sizeof(int), hipMemcpyDeviceToHost));
                tiedAny |= t;
            }
            whole = tiedAny != 0;
            m->lastPath = whole ? 2 : 1;
        } else {
            m->lastPath = 2;
        }
        if (whole) M_TRY(m, gather_packed(m, perDev));
        // Every device then sends ITS share of the (identical) merged tables home.
        M_TRY(m, run_workers([&](int g) {
            Dev &d = m->dev[g];
            W_HIP(d, hipSetDevice(d.id));
            if (whole) {
                kb::MergeParams mp;
                memset(&mp, 0, sizeof(mp));
                mp.gain = d.packed + sl.offGain;
                mp.row4col = d.packed + sl.offR4C;
                mp.nf = d.packed + sl.offNf;
                mp.shardStride = (long long)perDev;
                mp.spd = spd;
                mp.nShard = G * spd;
                mp.k = k;
                mp.maxCol = maxCol;
                mp.ldCol = maxCol;
                mp.maximize = opts->maximize;
                mp.outGain = d.mGain;
                mp.outRow4col = d.mR4C;
                mp.outNf = d.mNf;
                mp.inI8 = esz == 1 ? 1 : 0;
                if (gainsFirst) {  // (the first pass wrote into these: the merge expects them as the memsets left them)
                    W_HIP(d, hipMemsetAsync(d.mR4C, 0xFF, (size_t)B * k * maxCol * 4, d.stream));
                    W_HIP(d, hipMemsetAsync(d.mGain, 0, (size_t)B * k * 8, d.stream));
                }
                const hipError_t e = kb::launch_merge_topk(mp, B, d.stream);
                if (e != hipSuccess) { wfail(d, KBEST_ERR_HIP, std::string("merge kernel launch: ") + hipGetErrorString(e)); return; }
            } else {
                // the merged rows as the caller's int32 table, -1 / 0.0 in the slots beyond the number found
                hipError_t e = kb::launch_widen_i8(d.mRow8, d.mR4C, (long long)nRows8, d.stream);
                if (e == hipSuccess) e = kb::launch_fill_unused(d.mNf, nullptr, nullptr, B, k, maxCol, maxRow, d.mR4C, nullptr, d.mGain, false, d.stream);
                if (e != hipSuccess) { wfail(d, KBEST_ERR_HIP, std::string("widen / fill kernel launch: ") + hipGetErrorString(e)); return; }
            }
            const int b0 = (int)((long long)B * g / G), nb = (int)((long long)B * (g + 1) / G) - b0;
            W_HIP(d, hipStreamSynchronize(d.stream));
            d.issued = false;
            if (nb > 0) {
                W_HIP(d, hipMemcpy(gain + (size_t)b0 * k, d.mGain + (size_t)b0 * k, (size_t)nb * k * 8, hipMemcpyDeviceToHost));
                W_HIP(d, hipMemcpy(row4col + (size_t)b0 * k * maxCol, d.mR4C + (size_t)b0 * k * maxCol, (size_t)nb * k * maxCol * 4, hipMemcpyDeviceToHost));
                W_HIP(d, hipMemcpy(nf + b0, d.mNf + b0, (size_t)nb * 4, hipMemcpyDeviceToHost));
            }
            d.t[5] = kb::now_s() - m->t0;
        }));
        if (col4row)  // not part of the exchange: the inverse of row4col, rows without a real column -1
            for (int b = 0; b < B; b++)
                for (int s = 0; s < k; s++) {
                    int32_t *c = col4row + ((size_t)b * k + s) * maxRow;
                    for (int r = 0; r < maxRow; r++) c[r] = -1;
                    if (s >= nf[b]) continue;
                    const int M = nCol ? nCol[b] : maxCol;
                    const int32_t *r4 = row4col + ((size_t)b * k + s) * maxCol;
                    for (int j = 0; j < M; j++)
                        if (r4[j] >= 0 && r4[j] < maxRow) c[r4[j]] = j;
                }
    }
#undef W_HIP
#undef W_TRY
    for (int b = 0; b < B; b++)
        if (nf[b] < 0) return mfail(m, nf[b] == -1 ? KBEST_ERR_UNSUPPORTED : KBEST_ERR_INTERNAL, "kbest_batch_f64_multi: a problem came back with nf < 0");
    return KBEST_OK;
}

// Host times of the last kbest_batch_f64_multi[_ex] call, per device, in seconds since the call was entered:
// [0] the device's worker started, [1] its first upload was issued, [2] its first kernel was issued, [3] it was fed (batch mode:
// its own results are back in the caller's tables as well), [4] the exchange was issued, [5] everything of the call is done.
int kbest_multi_timeline(const kbest_multi *m, double *out, int capDevices)
{
    if (!m || !out) return KBEST_ERR_BAD_ARG;
    const int n = (int)m->dev.size() < capDevices ? (int)m->dev.size() : capDevices;
    for (int g = 0; g < n; g++)
        for (int i = 0; i < KBEST_MULTI_STAMPS; i++) out[(size_t)g * KBEST_MULTI_STAMPS + i] = m->dev[g].t[i];
    return n;
}

int kbest_batch_f64_multi(kbest_multi *m, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow,
                          const int32_t *nCol, const double *cost, int k, int32_t *row4col, int32_t *col4row, double *gain,
                          int32_t *nf)
{
    return kbest_batch_f64_multi_ex(m, opts, KBEST_MULTI_BATCH, 0, B, maxRow, maxCol, nRow, nCol, cost, k, row4col, col4row, gain, nf);
}

int kbest_multi_last_tie_flags(kbest_multi *m, int32_t *flags, int cap)
{
    if (!m || (cap > 0 && !flags)) return KBEST_ERR_BAD_ARG;
    const int n = (int)m->lastTie.size();
    for (int i = 0; i < n && i < cap; i++) flags[i] = m->lastTie[i];
    return n;
}

long long kbest_multi_exchange_bytes(const kbest_multi *m, int *path)
{
    if (!m) return -1;
    if (path) *path = m->lastPath;
    return (long long)m->lastSent;
}

int kbest_multi_tables_agree(kbest_multi *m)
{
    if (!m || m->lastB == 0) return KBEST_ERR_BAD_ARG;
    for (auto &d : m->dev) d.issued = false;
    if (m->lastMode == KBEST_MULTI_BATCH) {
        std::vector<unsigned char> a(m->lastBytes), b(m->lastBytes);
        for (size_t g = 0; g < m->dev.size(); g++) {
            Dev &d = m->dev[g];
            M_HIP(m, hipSetDevice(d.id));
            M_HIP(m, hipMemcpy(g ? b.data() : a.data(), d.packed, m->lastBytes, hipMemcpyDeviceToHost));
            if (g && memcmp(a.data(), b.data(), m->lastBytes)) return 0;
        }
        return 1;
    }
    const size_t nG = (size_t)m->lastB * m->lastK, nR = nG * m->lastCol;
    std::vector<double> g0(nG), g1(nG);
    std::vector<int32_t> r0(nR), r1(nR), n0(m->lastB), n1(m->lastB);
    for (size_t g = 0; g < m->dev.size(); g++) {
        Dev &d = m->dev[g];
        M_HIP(m, hipSetDevice(d.id));
        M_HIP(m, hipMemcpy(g ? g1.data() : g0.data(), d.mGain, nG * 8, hipMemcpyDeviceToHost));
        M_HIP(m, hipMemcpy(g ? r1.data() : r0.data(), d.mR4C, nR * 4, hipMemcpyDeviceToHost));
        M_HIP(m, hipMemcpy(g ? n1.data() : n0.data(), d.mNf, (size_t)m->lastB * 4, hipMemcpyDeviceToHost));
        if (g && (memcmp(g0.data(), g1.data(), nG * 8) || memcmp(r0.data(), r1.data(), nR * 4) ||
                  memcmp(n0.data(), n1.data(), (size_t)m->lastB * 4)))
            return 0;
    }
    return 1;
}

}  // extern "C"
