// kbest_merge.hip -- the "global k-best heap" of the subtree-sharded enumeration (north star: "RCCL allgather of
// per-rank top-k costs into a global k-best heap"), on the device.
//
// Murty's partition of the root is disjoint (split, shortestPathCPP.cpp:455-532): shard s of S enumerates only the root's
// children on columns c with c % S == s and produces its own k best -- slot 0 is the root itself on every shard, slots 1..
// are that shard's subtrees in increasing cost (decreasing profit when maximising).  The global k best are the root
// followed by the k - 1 best of the union of the shards' slots 1..: a k-way merge of S sorted lists.  One workgroup per
// problem; every candidate finds its own output position by counting the candidates that precede it -- per shard a binary
// search on the gain, then a walk over the run of equal gains, which are ordered by the assignment itself (lexicographic
// row4col) so that the merged table does not depend on the number of shards or on which shard a hypothesis came from
// (exact ties: SURVEY 8(a) quirk 7; the reference's own order there is a heap artefact).
#include <hip/hip_runtime.h>

#include "kbest_engine.h"
#include "kbest_ties.h"

namespace kb {

namespace {

// (gain, assignment) of candidate A strictly before candidate B in the merged order?
template <typename T> __device__ __forceinline__ bool before(double ga, const T *ra, double gb, const T *rb, int M, bool maximize)
{
    if (ga != gb) return maximize ? (ga > gb) : (ga < gb);
    for (int c = 0; c < M; c++)
        if (ra[c] != rb[c]) return ra[c] < rb[c];
    return false;
}

}  // namespace

// T: the type of the SHARDS' row4col tables (int32, or int8 when every index fits a byte: the multi-device exchange); the merged
// table is int32 either way.
template <typename T> __global__ void __launch_bounds__(256) merge_topk_kernel(MergeParams p)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int S = p.nShard, k = p.k, M = p.maxCol;
    const bool maximize = p.maximize != 0;
    // shard s: block s / spd (blocks shardStride bytes apart: the devices' slices), table s % spd inside the block (tables of
    // gridDim.x problems back to back); spd <= 1: one shard per block, as the packed per-rank slices are
    const int spd = p.spd > 1 ? p.spd : 1;
    const long long nB = gridDim.x;
    auto gainOf = [&](int s) { return reinterpret_cast<const double *>(p.gain + (long long)(s / spd) * p.shardStride) + ((long long)(s % spd) * nB + b) * k; };
    const long long sR = p.strideR4C ? p.strideR4C : p.shardStride, sN = p.strideNf ? p.strideNf : p.shardStride;
    auto rowsOf = [&](int s) { return reinterpret_cast<const T *>(p.row4col + (long long)(s / spd) * sR) + ((long long)(s % spd) * nB + b) * k * p.ldCol; };
    auto nfOf = [&](int s) { return reinterpret_cast<const int *>(p.nf + (long long)(s / spd) * sN)[(long long)(s % spd) * nB + b]; };
    double *og = p.outGain + (long long)b * k;
    int *orow = p.outRow4col + (long long)b * k * p.ldCol;
    const int nf0 = nfOf(0);
    if (nf0 <= 0) {  // infeasible (every shard solves the same root): kBest2D returns 0 (cpp:588-593); errors pass through
        if (tid == 0) p.outNf[b] = nf0;
        return;
    }
    __shared__ int total;
    if (tid == 0) total = 0;
    __syncthreads();
    // the root: slot 0 of shard 0
    if (tid == 0) og[0] = gainOf(0)[0];
    for (int c = tid; c < M; c += 256) orow[c] = (int)rowsOf(0)[c];
    int mine = 0;
    for (int idx = tid; idx < S * k; idx += 256) {
        const int s = idx / k, i = idx - s * k;
        const int n = nfOf(s);
        if (i < 1 || i >= n) continue;  // the root, or beyond what the shard found
        mine++;
        const double g = gainOf(s)[i];
        const T *r = rowsOf(s) + (long long)i * p.ldCol;
        int pos = 0;
        for (int t = 0; t < S; t++) {
            const double *gt = gainOf(t);
            const int nt = nfOf(t);
            // entries 1 .. nt-1 of shard t are sorted by gain: the first one that is not strictly better than g
            int lo = 1, hi = nt > 1 ? nt : 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const bool better = maximize ? (gt[mid] > g) : (gt[mid] < g);
                if (better) lo = mid + 1; else hi = mid;
            }
            pos += lo - 1;
            const T *rt = rowsOf(t);
            for (int j = lo; j < nt && gt[j] == g; j++)  // the run of equal gains: ordered by the assignment
                pos += before(gt[j], rt + (long long)j * p.ldCol, g, r, M, maximize) ? 1 : 0;
        }
        if (1 + pos < k) {
            og[1 + pos] = g;
            for (int c = 0; c < M; c++) orow[(long long)(1 + pos) * p.ldCol + c] = (int)r[c];
        }
    }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    const int nOut = (1 + total < k) ? 1 + total : k;
    if (tid == 0) p.outNf[b] = nOut;
    if (p.outCol4row) {  // the inverse of the merged row4col (the writes above are this workgroup's own: visible after the barrier)
        int *oc = p.outCol4row + (long long)b * k * p.ldRow;
        for (int i = tid; i < nOut * p.ldRow; i += 256) oc[i] = -1;
        __syncthreads();
        for (int i = tid; i < nOut * M; i += 256) {
            const int s = i / M, c = i - s * M;
            const int r = orow[(long long)s * p.ldCol + c];
            if (r >= 0 && r < p.ldRow) oc[(long long)s * p.ldRow + r] = c;
        }
    }
}

// Output slots beyond the number found are never written by the enumeration kernels: give them the values the host
// entries promise (row4col / col4row -1, gain 0) -- one workgroup per problem.  In a ragged batch (nRow / nCol given) the
// kernels write only row4col[.., c < nCol[b]] and col4row[.., r < nRow[b]] of an emitted slot: the padding of those slots,
// columns [nCol[b], ldCol) and rows [nRow[b], ldRow), is -1 as well, so that the whole table is defined whatever the
// buffer held before (recycled device blocks, the caller's registered memory).
template <typename T>
__global__ void __launch_bounds__(256) fill_unused_kernel(const int *nf, const int *nRow, const int *nCol, int k, int ldCol, int ldRow,
                                                          T *row4col, T *col4row, double *gain)
{
    const int b = blockIdx.x;
    int n = nf[b];
    n = n < 0 ? 0 : (n > k ? k : n);
    const long long base = (long long)b * k;
    if (nRow) {  // the padding of the emitted slots
        const int M = nCol[b] < ldCol ? (nCol[b] > 0 ? nCol[b] : 0) : ldCol, N = nRow[b] < ldRow ? (nRow[b] > 0 ? nRow[b] : 0) : ldRow;
        const int padC = ldCol - M, padR = ldRow - N;
        for (int i = threadIdx.x; i < n * padC; i += 256) row4col[(base + i / padC) * ldCol + M + i % padC] = -1;
        if (col4row)
            for (int i = threadIdx.x; i < n * padR; i += 256) col4row[(base + i / padR) * ldRow + N + i % padR] = -1;
    }
    if (n >= k) return;
    for (int i = threadIdx.x + n * ldCol; i < k * ldCol; i += 256) row4col[base * ldCol + i] = -1;
    if (col4row)
        for (int i = threadIdx.x + n * ldRow; i < k * ldRow; i += 256) col4row[base * ldRow + i] = -1;
    for (int i = threadIdx.x + n; i < k; i += 256) gain[base + i] = 0.0;
}

// The launch behind every enumeration launch: one workgroup per problem.  Wave 0 brings runs of equal gains into the one order
// all kernels share -- (gain, row4col lexicographic), kbest_ties.h -- and reports the problem's KBEST_TIE_* flags (tieGain: the
// gain of the solution behind the tables, which the kernels enumerate for this purpose); the other waves define the unused
// slots and the padding of ragged batches as fill_unused_kernel does (FILL).  The two touch different entries of the tables.
template <typename T, bool FILL>
__global__ void __launch_bounds__(256) finish_tables_kernel(const int *nf, const int *nRow, const int *nCol, int k, int ldCol, int ldRow,
                                                            T *row4col, T *col4row, double *gain, const double *tieGain, int *tieFlags, int baseFlags,
                                                            int order)
{
    __shared__ unsigned short scr[TIE_SCR_U16];
    const int b = blockIdx.x;
    const long long base = (long long)b * k;
    // (the tie-free problem's whole cost is ONE round trip to memory: the gains are read before the count is known -- slots
    //  beyond it are allocated, just not meaningful -- and compared under the count)
    bool anyEq = false;
    if (threadIdx.x < 64 && !FILL)
        for (int s = threadIdx.x; s + 1 < k; s += 64) anyEq = anyEq || (gain[base + s] == gain[base + s + 1] && s + 1 < nf[b]);
    int n = nf[b];
    n = n < 0 ? 0 : (n > k ? k : n);
    const int M = nCol ? (nCol[b] < ldCol ? (nCol[b] > 0 ? nCol[b] : 0) : ldCol) : ldCol;
    const int N = nRow ? (nRow[b] < ldRow ? (nRow[b] > 0 ? nRow[b] : 0) : ldRow) : ldRow;
    if (threadIdx.x < 64) {
        const double extra = tieGain ? tieGain[b] : __longlong_as_double(0x7ff8000000000000LL);
        if (!FILL && __ballot(anyEq) == 0ull) {  // no two equal gains inside: only the slot behind the tables can tie
            if (tieFlags && threadIdx.x == 0) tieFlags[b] = baseFlags | ((n == k && n > 0 && gain[base + n - 1] == extra) ? KBEST_TIE_BOUNDARY : 0);
            return;
        }
        const int fl = tie_tail(gain + base, reinterpret_cast<int *>(row4col), base * ldCol, reinterpret_cast<int *>(col4row), base * ldRow, n, M, N,
                                ldCol, ldRow, sizeof(T) == 1, scr, TIE_RUN_CAP, n == k && extra == extra, extra, order != 0);
        if (tieFlags && threadIdx.x == 0) tieFlags[b] = fl | baseFlags;
        if (FILL) return;
    }
    if (!FILL) return;
    const int t = threadIdx.x - 64;
    if (nRow) {  // the padding of the emitted slots
        const int padC = ldCol - M, padR = ldRow - N;
        for (int i = t; i < n * padC; i += 192) row4col[(base + i / padC) * ldCol + M + i % padC] = -1;
        if (col4row)
            for (int i = t; i < n * padR; i += 192) col4row[(base + i / padR) * ldRow + N + i % padR] = -1;
    }
    if (n >= k) return;
    for (int i = t + n * ldCol; i < k * ldCol; i += 192) row4col[base * ldCol + i] = -1;
    if (col4row)
        for (int i = t + n * ldRow; i < k * ldRow; i += 192) col4row[base * ldRow + i] = -1;
    for (int i = t + n; i < k; i += 192) gain[base + i] = 0.0;
}

hipError_t launch_finish_tables(const int *nf, const int *nRow, const int *nCol, int B, int k, int ldCol, int ldRow, int *row4col, int *col4row,
                                double *gain, bool tablesI8, const double *tieGain, int *tieFlags, bool fill, hipStream_t stream, int baseFlags, bool order)
{
    const int ord = order ? 1 : 0;
    if (B <= 0) return hipSuccess;
    signed char *r8 = reinterpret_cast<signed char *>(row4col), *c8 = reinterpret_cast<signed char *>(col4row);
    const dim3 g(B), bl(fill ? 256 : 64);
    if (tablesI8) {
        if (fill) hipLaunchKernelGGL((finish_tables_kernel<signed char, true>), g, bl, 0, stream, nf, nRow, nCol, k, ldCol, ldRow, r8, c8, gain, tieGain, tieFlags, baseFlags, ord);
        else hipLaunchKernelGGL((finish_tables_kernel<signed char, false>), g, bl, 0, stream, nf, nRow, nCol, k, ldCol, ldRow, r8, c8, gain, tieGain, tieFlags, baseFlags, ord);
    } else {
        if (fill) hipLaunchKernelGGL((finish_tables_kernel<int, true>), g, bl, 0, stream, nf, nRow, nCol, k, ldCol, ldRow, row4col, col4row, gain, tieGain, tieFlags, baseFlags, ord);
        else hipLaunchKernelGGL((finish_tables_kernel<int, false>), g, bl, 0, stream, nf, nRow, nCol, k, ldCol, ldRow, row4col, col4row, gain, tieGain, tieFlags, baseFlags, ord);
    }
    return hipGetLastError();
}

hipError_t launch_fill_unused(const int *nf, const int *nRow, const int *nCol, int B, int k, int ldCol, int ldRow, int *row4col, int *col4row,
                              double *gain, bool tablesI8, hipStream_t stream)
{
    if (tablesI8)  // KBEST_FLAG_TABLES_I8: the same tables, one byte per entry
        hipLaunchKernelGGL(fill_unused_kernel<signed char>, dim3(B), dim3(256), 0, stream, nf, nRow, nCol, k, ldCol, ldRow,
                           reinterpret_cast<signed char *>(row4col), reinterpret_cast<signed char *>(col4row), gain);
    else
        hipLaunchKernelGGL(fill_unused_kernel<int>, dim3(B), dim3(256), 0, stream, nf, nRow, nCol, k, ldCol, ldRow, row4col, col4row, gain);
    return hipGetLastError();
}

// int8 table -> int32 table (the multi-device entry's narrow staging: the kernels write a block's row4col as bytes -- that is what
// crosses PCIe --, the device's slice of the global table holds int32 as the exchange promises)
__global__ void __launch_bounds__(256) widen_i8_kernel(const signed char *src, int *dst, long long n)
{
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    // (a piece of the batch starts at b0 * k * maxCol entries: with an odd product neither pointer need be vector-aligned)
    if (i0 + 4 <= n && (reinterpret_cast<uintptr_t>(src + i0) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst + i0) & 15) == 0) {
        const char4 v = *reinterpret_cast<const char4 *>(src + i0);
        *reinterpret_cast<int4 *>(dst + i0) = make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
    } else {
        for (long long i = i0; i < n && i < i0 + 4; i++) dst[i] = (int)src[i];
    }
}

hipError_t launch_widen_i8(const signed char *src, int *dst, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(widen_i8_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

// Plain copy by a kernel (16 bytes per thread and pass; n a multiple of 4): the multi-device entry's narrow staging sends a
// piece's byte table, gains and counts home with it -- stores into host-mapped memory from the piece's own stream, no copy
// engine and no runtime call per table (asynchronous copies on the pieces' streams serialised the pieces under one HIP runtime:
// 1 024 x 64x64 through one logical device 4.4 ms instead of 2.8).
__global__ void __launch_bounds__(256) copy_words_kernel(const unsigned *src, unsigned *dst, long long nWords)
{
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < nWords; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= nWords && ((reinterpret_cast<uintptr_t>(src + i) | reinterpret_cast<uintptr_t>(dst + i)) & 15) == 0)
            *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
        else
            for (long long j = i; j < nWords && j < i + 4; j++) dst[j] = src[j];
    }
}

hipError_t launch_copy_words(const void *src, void *dst, long long bytes, hipStream_t stream)
{
    const long long nWords = bytes / 4;
    if (nWords <= 0) return hipSuccess;
    long long blocks = (nWords + 1023) / 1024;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const unsigned *>(src), static_cast<unsigned *>(dst), nWords);
    return hipGetLastError();
}

// words back to zero BY A KERNEL (stores through L2, where words that were updated by atomics live: a memset node did not reach
// them, kbest_engine.hip) -- the relay's progress words after a launch that failed
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned *w, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        __hip_atomic_store(w + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_zero_words(unsigned *w, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, n);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const MergeParams &p, int B, hipStream_t stream)
{
    if (p.inI8) hipLaunchKernelGGL(merge_topk_kernel<signed char>, dim3(B), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(merge_topk_kernel<int>, dim3(B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// The same merge from the shards' GAINS alone -- the north star's "allgather of per-rank top-k costs into a global k-best heap"
// (split, shortestPathCPP.cpp:455-532).  Every device holds all shards' gain[k] and nf (8 k + 4 bytes per matrix and
// shard: what the all-gather moved) but only ITS OWN shards' row4col lists.  One workgroup per matrix: every candidate's merged
// position follows from the gains (per shard a binary search); the merged gains and the count are written in full -- identical on
// every device --, and the rows of the device's own winners are scattered into a byte table that is zero elsewhere, so that ONE
// sum all-reduce of that table (k * M bytes per matrix, whatever the number of shards) completes it everywhere.  Two candidates
// with EXACTLY the same gain cannot be ordered without their assignments (the rule is (gain, row4col lexicographic), kbest_c.h):
// such a matrix is flagged in `tied` and the caller falls back to the exchange of the whole lists for the call.
__global__ void __launch_bounds__(256) merge_gains_kernel(MergeGainsParams p)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int S = p.nShard, k = p.k, M = p.maxCol;
    const bool maximize = p.maximize != 0;
    // shard s: block s / spd of the gathered heads (blockStride bytes apart), table s % spd inside it
    const int spd = p.spd > 1 ? p.spd : 1;
    auto gainOf = [&](int s) { return reinterpret_cast<const double *>(p.gain + (long long)(s / spd) * p.blockStride) + ((long long)(s % spd) * p.B + b) * k; };
    auto nfOf = [&](int s) { return reinterpret_cast<const int *>(p.nf + (long long)(s / spd) * p.blockStride)[(long long)(s % spd) * p.B + b]; };
    double *og = p.outGain + (long long)b * k;
    signed char *orow = p.outRow8 + (long long)b * k * M;
    const int nf0 = nfOf(0);
    if (nf0 <= 0) {  // infeasible (every shard solves the same root): kBest2D returns 0 (cpp:588-593); errors pass through
        if (tid == 0) p.outNf[b] = nf0;
        return;
    }
    __shared__ int total, anyTie;
    if (tid == 0) { total = 0; anyTie = 0; }
    __syncthreads();
    // the root: slot 0 of shard 0 (its row comes from the device that holds shard 0)
    if (tid == 0) og[0] = gainOf(0)[0];
    if (p.ownLo == 0 && p.ownHi > 0)
        for (int c = tid; c < M; c += 256) orow[c] = p.ownRow8[(long long)b * k * M + c];
    int mine = 0, tie = 0;
    for (int idx = tid; idx < S * k; idx += 256) {
        const int s = idx / k, i = idx - s * k;
        const int n = nfOf(s);
        if (i < 1 || i >= n) continue;  // the root, or beyond what the shard found
        mine++;
        const double g = gainOf(s)[i];
        int pos = 0, equal = 0;
        for (int t = 0; t < S; t++) {
            const double *gt = gainOf(t);
            const int nt = nfOf(t);
            // entries 1 .. nt-1 of shard t are sorted by gain: [lo, up) is the run equal to g
            int lo = 1, hi = nt > 1 ? nt : 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const bool better = maximize ? (gt[mid] > g) : (gt[mid] < g);
                if (better) lo = mid + 1; else hi = mid;
            }
            int up = lo;
            while (up < nt && gt[up] == g) up++;
            equal += up - lo;
            // (an order of the run by (shard, slot): only its SIZE matters -- a run of more than one is what the flag reports)
            pos += (lo - 1) + (t < s ? up - lo : (t == s ? i - lo : 0));
        }
        // a winner, or the first loser (the candidate that would take slot k): an equal gain next to it leaves the order -- or the
        // members of the level at slot k -- to the assignments
        if (equal > 1 && 1 + pos <= k) tie = 1;
        if (1 + pos < k) {
            og[1 + pos] = g;
            if (s >= p.ownLo && s < p.ownHi) {
                const signed char *r = p.ownRow8 + (((long long)(s - p.ownLo) * p.B + b) * k + i) * M;
                signed char *o = orow + (long long)(1 + pos) * M;
                for (int c = 0; c < M; c++) o[c] = r[c];
            }
        }
    }
    if (mine) atomicAdd(&total, mine);
    if (tie) atomicOr(&anyTie, 1);
    __syncthreads();
    if (tid == 0) {
        p.outNf[b] = (1 + total < k) ? 1 + total : k;
        if (anyTie) atomicOr(p.tied, 1);
    }
}

hipError_t launch_merge_gains(const MergeGainsParams &p, hipStream_t stream)
{
    if (p.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(merge_gains_kernel, dim3(p.B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// int32 table -> int8 table (the multi-device entry: a block's row4col staged as int32 for the caller goes into the device's slice of
// the exchange as bytes), and dst += src on byte tables (logical devices on one GPU: the sum all-reduce of the winners' rows done by hand)
__global__ void __launch_bounds__(256) narrow_i32_kernel(const int *src, signed char *dst, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (signed char)src[i];
}

hipError_t launch_narrow_i32(const int *src, signed char *dst, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(narrow_i32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) add_i8_kernel(signed char *dst, const signed char *src, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (signed char)(dst[i] + src[i]);
}

hipError_t launch_add_i8(signed char *dst, const signed char *src, long long n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(add_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

}  // namespace kb
