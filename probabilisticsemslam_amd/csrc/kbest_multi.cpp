// kbest_multi.cpp -- the multi-device entry points of include/kbest_c.h: one engine context and one stream per GPU,
// contiguous block sharding of the batch, and the RCCL all-gather (over xGMI) of the packed per-device result tables
// (gain[k], row4col[k*M], nf per matrix -- SURVEY 8(e)) that leaves every device with the same global k-best table.
// This is the C++ side of BASELINE.json's config 4 ("sharded across 8 MI355X via RCCL top-k allgather"): a host
// program written like the reference (one process, plain C++) shards without a Python launcher.
//
// The matrices are independent, so there is NO data-path collective: each device solves its block with the same
// kernels as the single-device entries, writing straight into its slice of the global table; the all-gather is
// in place (send buffer = own slice of the receive buffer) and is the only exchange.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library stays loadable -- and every single-device entry
// usable -- on a host without RCCL, and a process that already carries another copy of RCCL (PyTorch bundles one) does
// not get a second one forced into its link map unless it asks for the multi-device entries.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "kbest_c.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
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
    // device buffers, grown on demand
    double *cost = nullptr;
    int32_t *shape = nullptr;   // nRow | nCol of this device's block
    int32_t *c4r = nullptr;     // col4row of this device's block (not gathered: SURVEY 8(e) exchanges gain, row4col, nf)
    int32_t *gR4C = nullptr;    // global tables, identical on every device after the gather
    double *gGain = nullptr;
    int32_t *gNf = nullptr;
    size_t costB = 0, shapeB = 0, c4rB = 0, gR4CB = 0, gGainB = 0, gNfB = 0;
};

}  // namespace

struct kbest_multi {
    std::vector<Dev> dev;
    Rccl rccl;
    std::string err;
    // shape of the last call (for kbest_multi_tables_agree)
    int lastB = 0, lastPad = 0, lastK = 0, lastCol = 0;
};

namespace {

int mfail(kbest_multi *m, int code, const std::string &what)
{
    if (m) m->err = what;
    return code;
}

#define M_HIP(m, call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return mfail(m, KBEST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define M_NCCL(m, call)                                                                                   \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return mfail(m, KBEST_ERR_HIP, std::string(#call) + ": " + m->rccl.GetErrorString(r_)); \
    } while (0)

template <class T> int grow(kbest_multi *m, T *&p, size_t &have, size_t need)
{
    if (need <= have) return KBEST_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    have = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), need);
    if (e != hipSuccess) return mfail(m, KBEST_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    have = need;
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
    std::string err;
    if (!m->rccl.load(err)) { kbest_destroy_multi(m); return KBEST_ERR_NO_DEVICE; }
    std::vector<ncclComm_t> comms(nDev);
    if (m->rccl.CommInitAll(comms.data(), nDev, device_ids) != ncclSuccess) { kbest_destroy_multi(m); return KBEST_ERR_HIP; }
    for (int g = 0; g < nDev; g++) m->dev[g].comm = comms[g];
    *out = m;
    return KBEST_OK;
}

int kbest_destroy_multi(kbest_multi *m)
{
    if (!m) return KBEST_OK;
    for (auto &d : m->dev) {
        (void)hipSetDevice(d.id);
        if (d.comm && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(d.comm);
        for (void *p : {(void *)d.cost, (void *)d.shape, (void *)d.c4r, (void *)d.gR4C, (void *)d.gGain, (void *)d.gNf})
            if (p) (void)hipFree(p);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.ctx) kbest_destroy(d.ctx);
    }
    delete m;
    return KBEST_OK;
}

int kbest_multi_size(const kbest_multi *m) { return m ? (int)m->dev.size() : 0; }

const char *kbest_multi_last_error(const kbest_multi *m) { return m ? m->err.c_str() : "null context"; }

int kbest_batch_f64_multi(kbest_multi *m, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow,
                          const int32_t *nCol, const double *cost, int k, int32_t *row4col, int32_t *col4row, double *gain,
                          int32_t *nf)
{
    if (!m) return KBEST_ERR_BAD_ARG;
    if (!opts || B < 0 || k < 1 || maxCol < 1 || maxRow < maxCol || !cost || !row4col || !gain || !nf ||
        (nRow == nullptr) != (nCol == nullptr))
        return mfail(m, KBEST_ERR_BAD_ARG, "kbest_batch_f64_multi: bad argument");
    if (B == 0) return KBEST_OK;
    const int G = (int)m->dev.size();
    const int pad = (B + G - 1) / G;  // matrices per device (the last devices may hold fewer): equal all-gather counts
    const size_t per = (size_t)maxRow * maxCol;
    m->lastB = B; m->lastPad = pad; m->lastK = k; m->lastCol = maxCol;
    // 1. every device: its block of cost matrices in, its slice of the global table solved in place
    for (int g = 0; g < G; g++) {
        Dev &d = m->dev[g];
        const int b0 = g * pad, nb = (b0 >= B) ? 0 : ((B - b0 < pad) ? B - b0 : pad);
        M_HIP(m, hipSetDevice(d.id));
        int rc = grow(m, d.gR4C, d.gR4CB, (size_t)G * pad * k * maxCol * 4);
        if (rc == KBEST_OK) rc = grow(m, d.gGain, d.gGainB, (size_t)G * pad * k * 8);
        if (rc == KBEST_OK) rc = grow(m, d.gNf, d.gNfB, (size_t)G * pad * 4);
        if (rc == KBEST_OK) rc = grow(m, d.cost, d.costB, (size_t)pad * per * 8);
        if (rc == KBEST_OK && col4row) rc = grow(m, d.c4r, d.c4rB, (size_t)pad * k * maxRow * 4);
        if (rc == KBEST_OK && nRow) rc = grow(m, d.shape, d.shapeB, (size_t)2 * pad * 4);
        if (rc != KBEST_OK) return rc;
        // slots the kernels do not write (beyond nf, padding problems) get defined values
        M_HIP(m, hipMemsetAsync(d.gR4C + (size_t)b0 * k * maxCol, 0xFF, (size_t)pad * k * maxCol * 4, d.stream));
        M_HIP(m, hipMemsetAsync(d.gGain + (size_t)b0 * k, 0, (size_t)pad * k * 8, d.stream));
        M_HIP(m, hipMemsetAsync(d.gNf + b0, 0, (size_t)pad * 4, d.stream));
        if (nb == 0) continue;
        if (col4row) M_HIP(m, hipMemsetAsync(d.c4r, 0xFF, (size_t)pad * k * maxRow * 4, d.stream));
        M_HIP(m, hipMemcpyAsync(d.cost, cost + (size_t)b0 * per, (size_t)nb * per * 8, hipMemcpyHostToDevice, d.stream));
        if (nRow) {
            M_HIP(m, hipMemcpyAsync(d.shape, nRow + b0, (size_t)nb * 4, hipMemcpyHostToDevice, d.stream));
            M_HIP(m, hipMemcpyAsync(d.shape + pad, nCol + b0, (size_t)nb * 4, hipMemcpyHostToDevice, d.stream));
        }
        rc = kbest_reserve(d.ctx, nb, maxRow, k);
        if (rc != KBEST_OK) return mfail(m, rc, std::string("device ") + std::to_string(d.id) + ": " + kbest_last_error(d.ctx));
        rc = kbest_batch_f64_dev(d.ctx, opts, nb, maxRow, maxCol, nRow ? d.shape : nullptr, nRow ? d.shape + pad : nullptr, d.cost,
                                 nullptr, k, d.gR4C + (size_t)b0 * k * maxCol, col4row ? d.c4r : nullptr,
                                 d.gGain + (size_t)b0 * k, d.gNf + b0, nullptr, d.stream);
        if (rc != KBEST_OK) return mfail(m, rc, std::string("device ") + std::to_string(d.id) + ": " + kbest_last_error(d.ctx));
    }
    // 2. the one exchange: in-place all-gather of (gain[k], row4col[k*M], nf) per matrix, stream-ordered behind each
    //    device's kernel
    M_NCCL(m, m->rccl.GroupStart());
    for (int g = 0; g < G; g++) {
        Dev &d = m->dev[g];
        const size_t b0 = (size_t)g * pad;
        M_NCCL(m, m->rccl.AllGather(d.gGain + b0 * k, d.gGain, (size_t)pad * k, ncclDouble, d.comm, d.stream));
        M_NCCL(m, m->rccl.AllGather(d.gR4C + b0 * k * maxCol, d.gR4C, (size_t)pad * k * maxCol, ncclInt32, d.comm, d.stream));
        M_NCCL(m, m->rccl.AllGather(d.gNf + b0, d.gNf, (size_t)pad, ncclInt32, d.comm, d.stream));
    }
    M_NCCL(m, m->rccl.GroupEnd());
    // 3. results: the global table from device 0 (any device holds it), col4row from the device that solved the block
    for (int g = 0; g < G; g++) {
        Dev &d = m->dev[g];
        M_HIP(m, hipSetDevice(d.id));
        M_HIP(m, hipStreamSynchronize(d.stream));
    }
    Dev &d0 = m->dev[0];
    M_HIP(m, hipSetDevice(d0.id));
    M_HIP(m, hipMemcpy(row4col, d0.gR4C, (size_t)B * k * maxCol * 4, hipMemcpyDeviceToHost));
    M_HIP(m, hipMemcpy(gain, d0.gGain, (size_t)B * k * 8, hipMemcpyDeviceToHost));
    M_HIP(m, hipMemcpy(nf, d0.gNf, (size_t)B * 4, hipMemcpyDeviceToHost));
    if (col4row) {
        for (int g = 0; g < G; g++) {
            Dev &d = m->dev[g];
            const int b0 = g * pad, nb = (b0 >= B) ? 0 : ((B - b0 < pad) ? B - b0 : pad);
            if (nb == 0) continue;
            M_HIP(m, hipSetDevice(d.id));
            M_HIP(m, hipMemcpy(col4row + (size_t)b0 * k * maxRow, d.c4r, (size_t)nb * k * maxRow * 4, hipMemcpyDeviceToHost));
        }
    }
    for (int b = 0; b < B; b++)
        if (nf[b] < 0) return mfail(m, nf[b] == -1 ? KBEST_ERR_UNSUPPORTED : KBEST_ERR_INTERNAL, "kbest_batch_f64_multi: a problem came back with nf < 0");
    return KBEST_OK;
}

int kbest_multi_tables_agree(kbest_multi *m)
{
    if (!m || m->lastB == 0) return KBEST_ERR_BAD_ARG;
    const size_t nG = (size_t)m->lastB * m->lastK, nR = nG * m->lastCol;
    std::vector<double> g0(nG), g1(nG);
    std::vector<int32_t> r0(nR), r1(nR), n0(m->lastB), n1(m->lastB);
    for (size_t g = 0; g < m->dev.size(); g++) {
        Dev &d = m->dev[g];
        M_HIP(m, hipSetDevice(d.id));
        M_HIP(m, hipMemcpy(g ? g1.data() : g0.data(), d.gGain, nG * 8, hipMemcpyDeviceToHost));
        M_HIP(m, hipMemcpy(g ? r1.data() : r0.data(), d.gR4C, nR * 4, hipMemcpyDeviceToHost));
        M_HIP(m, hipMemcpy(g ? n1.data() : n0.data(), d.gNf, (size_t)m->lastB * 4, hipMemcpyDeviceToHost));
        if (g && (memcmp(g0.data(), g1.data(), nG * 8) || memcmp(r0.data(), r1.data(), nR * 4) ||
                  memcmp(n0.data(), n1.data(), (size_t)m->lastB * 4)))
            return 0;
    }
    return 1;
}

}  // extern "C"
