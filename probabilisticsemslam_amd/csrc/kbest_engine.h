// kbest_engine.h -- internal interface between the C ABI (kbest_capi.cpp) and
// the device code (kbest_engine.hip).  Not installed; include/kbest_c.h is the
// public boundary.
#ifndef KBEST_ENGINE_H
#define KBEST_ENGINE_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kbest_c.h"

namespace kb {

typedef unsigned long long u64;
typedef unsigned int u32;

// Kernel arguments of one batched launch (all pointers are device pointers).
struct Params {
    const double *cost;       // packed column-major cost blocks
    const long long *costOff; // per-problem offset in doubles, or nullptr (uniform)
    const int *nRow;          // per-problem shapes, or nullptr (uniform maxRow x maxCol)
    const int *nCol;
    int maxRow, maxCol;       // shape bounds = leading dimensions of the outputs
    int k;
    int maximize, useCutoff;
    unsigned flags;
    double cutoff;
    int rootColOffset, rootColStride;
    int *row4col;             // [B][k][maxCol]
    int *col4row;             // [B][k][maxRow]
    double *gain;             // [B][k]
    int *nf;                  // [B]
    long long *pushed;        // [B] or nullptr
    unsigned char *states;    // workspace: [B][k] saved hypotheses, stateStride bytes each
    long long stateStride;
};

struct WeightParams {
    const int *nL, *nM;
    const double *cost;
    const long long *costOff;
    const double *gain;       // [B][k]
    const int *row4col;       // [B][k][maxCol]
    const int *nf;            // [B]
    double *probs;
    const long long *probOff;
    int k, maxCol;
};

// Bytes of one saved hypothesis: u[D] v[D] (fp64), row4col[D] col4row[D] (u8),
// forbidden-row mask, gain, activeCol.
__host__ __device__ inline long long state_stride(int maxRow)
{
    return (((long long)18 * maxRow + 7) & ~7LL) + 24;
}

// LDS carve-up of one workgroup (= one cost matrix).
struct Lds {
    int offC, offU, offV, offPrefix, offChildGain, offPoolG[2], offPoolM[2], offR4C, offC4R, offCtrl, total;
};

__host__ __device__ inline Lds lds_layout(int maxRow, int k)
{
    Lds L;
    const int ldc = maxRow | 1;
    int o = 0;
    L.offC = o;          o += maxRow * ldc * 8;  // shifted, zero-padded cost tile
    L.offU = o;          o += maxRow * 8;        // parent duals per column
    L.offV = o;          o += maxRow * 8;        // parent duals per row
    L.offPrefix = o;     o += maxRow * 8;        // parent's serial gain prefix sums
    L.offChildGain = o;  o += 64 * 8;            // gains of this sweep's children
    L.offPoolG[0] = o;   o += k * 8;             // candidate pool, ping
    L.offPoolG[1] = o;   o += k * 8;             //                 pong
    L.offPoolM[0] = o;   o += k * 4;
    L.offPoolM[1] = o;   o += k * 4;
    o = (o + 7) & ~7;
    L.offR4C = o;        o += maxRow * 4;
    L.offC4R = o;        o += maxRow * 4;
    o = (o + 7) & ~7;
    L.offCtrl = o;       o += 96;                // struct Ctrl
    L.total = (o + 15) & ~15;
    return L;
}

hipError_t launch_kbest(const Params &p, int B, int nWaves, hipStream_t stream);
hipError_t launch_weights(const WeightParams &p, int B, hipStream_t stream);

}  // namespace kb
#endif
