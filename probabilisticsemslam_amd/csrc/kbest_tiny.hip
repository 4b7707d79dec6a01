// kbest_tiny.hip -- MI355X (gfx950): the fused association path for frames with a HANDFUL of measurements.
//
// The reference's real frames hold 3-5 measurements (README.md:11; getAssignmentProbs, assignment.cpp:38-74, once per
// frame from system.cpp:268).  conditionCosts (assignment.cpp:439-525) leaves a (condL + nM) x nM problem whose
// assignments -- one row per measurement column, the rows left over on the zero-padded columns (shortestPathCPP.cpp:582-585)
// -- number N (N-1) ... (N-nM+1): 504 for six landmarks and three measurements, 55 440 for six and five.  Murty's
// enumeration (kBest2DCutoff, shortestPathCPP.cpp:646-733) finds the k cheapest of them one shortest-path search at a time --
// on this device a chain of rounds of ~15 us each, of which a 9 x 3 frame needs five.  When ALL assignments are that few,
// the k cheapest are found by looking at all of them, in two passes and one small sort, in a fraction of one such round:
//
//   * every assignment's gain is calcGain's sum (shortestPathCPP.cpp:59-80): the entries C[row4col[c]][c] added left to
//     right from 0.0 -- the same additions in the same order, so the same bits as the reference's gainBest (the zero
//     columns add +0.0, and x + 0.0 == x); an assignment through a +inf entry (beyond the gate, assignment.cpp:487-493) is
//     one the reference's searches never return (cpp:327);
//   * what kBest2DCutoff emits is the ascending run of the k smallest gains up to gainBest[0] + cutoff (cpp:705-719); what
//     assignmentProb makes of it (assignment.cpp:616-648) does not depend on the order of equal gains (equal gains have equal
//     weights, every probs[col][row] receives the same values in the same order);
//   * a greedy assignment bounds what can be emitted at all (greedy + cutoff; on large frames the k-th cheapest of its one- and
//     two-column neighbours, which are assignments too, bounds the k-th best gain more tightly); the prefixes (rows of the first nM - 1 or
//     nM - 2 columns) that stay below it and off the +inf entries are decoded ONCE into an LDS list;
//   * pass 1: the threads share the list's work items (prefix x free row of the next column; loop over the last column),
//     reduce the minimum and fill a 1 024-bucket histogram of the gains over [0, 42 nM] (conditioned entries lie in [0, 42]);
//     the first bucket at which the cumulated count reaches k bounds the k-th gain; pass 2 collects the assignments up to
//     that bucket (a few more than k), a rank sort orders them by (gain, index), the first min(k, those within the cutoff)
//     are the solutions; then the weights, exactly as in kbest_small.hip.
//
// A frame whose list overflows (more than TINY_CAP candidates up to the k-th bucket: thousands of equal gains) or that is
// not what the host promised comes back with nf = -2 and goes through the enumeration kernels like any other.
// fp64 add / compare / exp only; no fast-math.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "kbest_engine.h"
#include "kbest_wave.h"

namespace kb {

namespace {

constexpr double TN_GATE = 42.0;  // assignment.cpp:9
constexpr int TN_LDT = TINY_MAX_ROW + 1;
constexpr int TN_BUCKETS = 1024;

struct TCtrl {
    unsigned long long minBits;  // smallest gain (non-negative doubles order like their bits)
    int listN;                   // candidates collected
    int nWithin;                 // ... of which within the cutoff
    int bStar;                   // bucket of the k-th smallest gain
    int total;                   // feasible assignments seen
    double limit;                // greedy assignment + cutoff: nothing beyond it is ever emitted
    int nFeas;                   // feasible prefixes found (kept: the first tiny_prefix_cap)
    u64 greedyUsed;              // rows of the greedy assignment (0: there is none)
};
static_assert(sizeof(TCtrl) <= 48, "the LDS carve-up of kbest_tiny_kernel gives TCtrl 48 bytes");

// the q-th prefix (rows of the columns 0 .. D-1, lexicographic in "which of the still free rows"): rows packed one byte per
// column, the set of used rows; false: the prefix runs through a +inf entry
__device__ __forceinline__ bool tiny_prefix(const double *Cs, int N, int D, u32 q, u64 &rowsOut, u64 &usedOut, double &accOut)
{
    // digits, least significant = column D-1
    u32 dig[TINY_MAX_COL];
#pragma unroll
    for (int c = TINY_MAX_COL - 2; c >= 0; c--) {
        if (c < D) {
            const u32 radix = (u32)(N - c);
            const u32 t = q / radix;
            dig[c] = q - t * radix;
            q = t;
        } else
            dig[c] = 0;
    }
    u64 rows = 0ull, used = 0ull;
    double acc = 0.0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < TINY_MAX_COL - 1; c++) {
        if (c < D) {
            // the dig[c]-th free row
            u64 freeM = ~used & ((N >= 64) ? ~0ull : ((1ull << N) - 1ull));
            for (u32 i = 0; i < dig[c]; i++) freeM &= freeM - 1ull;
            const int r = __builtin_ctzll(freeM);
            used |= 1ull << r;
            rows |= (u64)r << (8 * c);
            const double x = Cs[r + c * TN_LDT];
            acc = acc + x;  // calcGain (cpp:59-80): left to right from 0.0
            ok = ok && (x < d_inf());
        }
    }
    rowsOut = rows;
    usedOut = used;
    accOut = acc;
    return ok;
}

// Every feasible assignment once: thread = prefixes of D columns (tid, tid + NT, ...), the last one (INNER = 1) or two (INNER = 2)
// columns in loops over the rows still free; f(gain, id) with id = prefix << 12 | row of column M-2 << 6 | row of column M-1
// (INNER = 1: prefix << 6 | row of column M-1).  The gain is calcGain's sum (cpp:59-80): left to right from 0.0.
// Entries are >= 0, partial sums only grow: a prefix beyond `limit` has no completion at or below it and is skipped.
template <int INNER, typename F>
__device__ __forceinline__ void tiny_walk(const double *Cs, int N, int M, u32 nPre, int tid, int nThreads, double limit, F f)
{
    const double INF = d_inf();
    const u64 rowsAll = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
    const double *Clast = Cs + (M - 1) * TN_LDT;
    for (u32 q = (u32)tid; q < nPre; q += (u32)nThreads) {
        u64 rows, used;
        double acc;
        if (!tiny_prefix(Cs, N, M - INNER, q, rows, used, acc)) continue;
        if (acc > limit) continue;
        if (INNER == 1) {
            for (u64 fm = rowsAll & ~used; fm; fm &= fm - 1ull) {
                const int r = __builtin_ctzll(fm);
                const double g = acc + Clast[r];
                if (g < INF) f(g, (q << 6) | (u32)r);
            }
        } else {
            const double *Cprev = Cs + (M - 2) * TN_LDT;
            for (u64 f1 = rowsAll & ~used; f1; f1 &= f1 - 1ull) {
                const int r1 = __builtin_ctzll(f1);
                const double acc1 = acc + Cprev[r1];
                if (!(acc1 < INF) || acc1 > limit) continue;
                for (u64 fm = rowsAll & ~used & ~(1ull << r1); fm; fm &= fm - 1ull) {
                    const int r = __builtin_ctzll(fm);
                    const double g = acc1 + Clast[r];
                    if (g < INF) f(g, (q << 12) | ((u32)r1 << 6) | (u32)r);
                }
            }
        }
    }
}

// The same walk over a COMPACTED list of the feasible prefixes (rows packed, partial sum, prefix number): work item = (prefix,
// which of its free rows the first inner column takes), so that the threads share what is left evenly -- on gated frames nine
// prefixes in ten run through a +inf entry, and in the walk above their lanes idle while the tenth loops.
template <int INNER, typename F>
__device__ __forceinline__ void tiny_walk_list(const double *Cs, int N, int M, const u64 *preRows, const double *preAcc, const u32 *preQ,
                                               int nFeas, int tid, int nThreads, double limit, F f)
{
    const double INF = d_inf();
    const u64 rowsAll = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
    const double *Clast = Cs + (M - 1) * TN_LDT;
    const int D = M - INNER;
    if (INNER == 1) {  // one column left: the prefix' own loop over its free rows (an item per row would re-find it by counting)
        for (u32 i = (u32)tid; i < (u32)nFeas; i += (u32)nThreads) {
            const u64 rows = preRows[i];
            const double acc = preAcc[i];
            const u32 q = preQ[i];
            u64 used = 0ull;
            for (int c = 0; c < D; c++) used |= 1ull << ((rows >> (8 * c)) & 63ull);
            for (u64 fm = rowsAll & ~used; fm; fm &= fm - 1ull) {
                const int r = __builtin_ctzll(fm);
                const double g = acc + Clast[r];
                if (g < INF) f(g, (q << 6) | (u32)r);
            }
        }
        return;
    }
    const u32 F1 = (u32)(N - D);  // rows still free behind a prefix
    const u32 nItems = (u32)nFeas * F1;
    for (u32 t = (u32)tid; t < nItems; t += (u32)nThreads) {
        const u32 i = t / F1, j = t - i * F1;
        const u64 rows = preRows[i];
        const double acc = preAcc[i];
        const u32 q = preQ[i];
        u64 used = 0ull;
        for (int c = 0; c < D; c++) used |= 1ull << ((rows >> (8 * c)) & 63ull);
        u64 fm1 = rowsAll & ~used;
        for (u32 x = 0; x < j; x++) fm1 &= fm1 - 1ull;
        const int r1 = __builtin_ctzll(fm1);
        const double acc1 = acc + Cs[r1 + (M - 2) * TN_LDT];
        if (!(acc1 < INF) || acc1 > limit) continue;
        for (u64 fm = rowsAll & ~used & ~(1ull << r1); fm; fm &= fm - 1ull) {
            const int r = __builtin_ctzll(fm);
            const double g = acc1 + Clast[r];
            if (g < INF) f(g, (q << 12) | ((u32)r1 << 6) | (u32)r);
        }
    }
}

}  // namespace

template <int NT>
__global__ void __launch_bounds__(NT) kbest_tiny_kernel(SmallParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWV = NT / 64;
    const double INF = d_inf();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int k = p.kTab;  // (the caller's k; p.k is what the ENUMERATION kernels enumerate: one more, kbest_ties.h)
    if (p.tieFlags && threadIdx.x == 0) p.tieFlags[blockIdx.x] = 0;
    const int M = p.imm ? p.immCol : p.nCol[b];
    const int NR = p.imm ? p.immRow : p.nRow[b];
    const int nLout = p.imm ? p.immL : p.nL[b];
    const double *Cg = p.cost + (p.costOff ? p.costOff[b] : 0);
    double *probOut = p.probs + (p.probOff ? p.probOff[b] : 0);
    // LDS: tile | raw block | column minima | control | kept rows | row index | histogram | list (gain, id) | rank | solutions (gain, id) | weights | row table
    int o = 0;
    double *Cs = reinterpret_cast<double *>(smem + o);       o += TINY_MAX_COL * TN_LDT * 8;
    double *stage = reinterpret_cast<double *>(smem + o);    o += TINY_MAX_COL * TINY_MAX_ROW * 8;
    double *colMin = reinterpret_cast<double *>(smem + o);   o += TINY_MAX_COL * 8;
    TCtrl *ctl = reinterpret_cast<TCtrl *>(smem + o);        o += 48;
    unsigned char *gRow = smem + o;                          o += 8;   // the greedy assignment's rows
    u64 *keepW = reinterpret_cast<u64 *>(smem + o);          o += 8;
    unsigned short *rowIdx = reinterpret_cast<unsigned short *>(smem + o);  o += TINY_MAX_ROW * 2;
    u32 *hist = reinterpret_cast<u32 *>(smem + o);           o += TN_BUCKETS * 4;
    double *listG = reinterpret_cast<double *>(smem + o);    o += TINY_CAP * 8;
    u32 *listI = reinterpret_cast<u32 *>(smem + o);          o += TINY_CAP * 4;
    int *rankA = reinterpret_cast<int *>(smem + o);          o += TINY_CAP * 4;
    double *solG = reinterpret_cast<double *>(smem + o);     o += k * 8;
    u32 *solI = reinterpret_cast<u32 *>(smem + o);           o += k * 4;
    o = (o + 7) & ~7;
    double *wts = reinterpret_cast<double *>(smem + o);      o += k * 8;
    unsigned char *rTab = smem + o;                          o += k * TINY_MAX_COL;  // [k][M]
    o = (o + 7) & ~7;
    constexpr int PCAP = tiny_prefix_cap(NT);  // feasible prefixes kept for the passes
    u64 *preRows = reinterpret_cast<u64 *>(smem + o);        o += PCAP * 8;
    double *preAcc = reinterpret_cast<double *>(smem + o);   o += PCAP * 8;
    u32 *preQ = reinterpret_cast<u32 *>(smem + o);

    auto signal_done = [&]() {  // (as in kbest_small.hip: the host polls this counter on one-frame calls)
        if (p.done) {
            __syncthreads();
            if (tid == 0) {
                __threadfence_system();
                __hip_atomic_fetch_add(p.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (M == 0 || NR == 0) {  // an empty frame (assignment.cpp:50-51)
        if (tid == 0) p.nf[b] = 0;
        signal_done();
        return;
    }
    if (M < 2 || M > TINY_MAX_COL || NR > TINY_MAX_ROW || NR < M || nLout + M != NR) {  // not what this kernel takes: the enumeration kernels
        if (tid == 0) p.nf[b] = -2;
        signal_done();
        return;
    }
    for (int i = tid; i < M * (nLout + 1); i += NT) probOut[i] = 0.0;
    // ---- conditionCosts (assignment.cpp:439-525), as in kbest_small.hip ------------------------------------------------
    for (int i = tid; i < NR * M; i += NT) stage[i] = Cg[i];
    for (int i = tid; i < TINY_MAX_COL * TN_LDT; i += NT) Cs[i] = INF;
    for (int i = tid; i < TN_BUCKETS; i += NT) hist[i] = 0u;
    if (tid == 0) {
        ctl->minBits = 0x7ff0000000000000ull;
        ctl->listN = 0;
        ctl->nWithin = 0;
        ctl->bStar = TN_BUCKETS - 1;
        ctl->total = 0;
        ctl->nFeas = 0;
    }
    __syncthreads();
    int N;
    if (p.condition) {
        for (int c = wave; c < M; c += NWV) {  // column minima (:450-458)
            double m = INF;
            for (int r = lane; r < NR; r += 64) m = min_keep(m, stage[c * NR + r]);
            m = wave_min_f64(m);
            if (lane == 0) colMin[c] = m;
        }
        __syncthreads();
        if (wave == 0) {  // a row is kept iff some entry is within 42 of its column's minimum (:462-474)
            bool good = false;
            if (lane < NR)
                for (int c = 0; c < M; c++) good = good | (stage[c * NR + lane] <= colMin[c] + TN_GATE);
            const u64 m = __ballot(good);
            if (lane == 0) *keepW = m;
        }
        __syncthreads();
        const u64 keep = *keepW;
        N = __popcll(keep);
        if (N < M) {  // undefined in the reference (size_t underflow at assignment.cpp:60); the enumeration kernels answer -2 as well
            if (tid == 0) p.nf[b] = -2;
            signal_done();
            return;
        }
        if (tid < NR && ((keep >> tid) & 1ull)) {  // kept rows compacted in order; entries cost - colMin, +inf beyond the gate (:476-496)
            const int nr = __popcll(keep & ((1ull << tid) - 1ull));
            rowIdx[nr] = (unsigned short)tid;
            for (int c = 0; c < M; c++) {
                const double x = stage[c * NR + tid];
                Cs[nr + c * TN_LDT] = (x <= colMin[c] + TN_GATE) ? (x - colMin[c]) : INF;
            }
        }
    } else {
        // assignmentProb on a block that is conditioned already (the reference's own call, assignment.cpp:58-62).  kBest2DCutoff
        // shifts the matrix by its smallest entry (makeCostMatrixSafe, cpp:534-569) and adds CDelta * numCol back to every gain
        // (cpp:583, 626-630): on a conditioned block that entry is an exact 0.0 -- every column holds one -- the shift is the
        // identity and the gains are calcGain's sums as they stand.  Any other block (a negative entry, no zero) is not this
        // kernel's: -2.
        N = NR;
        double mn = INF;
        for (int i = tid; i < NR * M; i += NT) {
            const double x = stage[i];
            mn = min_keep(mn, x);
            const int c = i / NR, r = i - c * NR;
            Cs[r + c * TN_LDT] = (x == x) ? x : INF;  // NaN: every comparison the reference makes with it is false, like +inf
        }
        mn = wave_min_f64(mn);
        if (lane == 0) {  // (bit patterns of non-negative doubles order like the values; a negative entry is flagged apart)
            if (mn < 0.0) atomicAdd(&ctl->total, 1);
            else atomicMin(&ctl->minBits, (unsigned long long)__double_as_longlong(mn));
        }
        if (tid < NR) rowIdx[tid] = (unsigned short)tid;
        __syncthreads();
        const bool notMine = ctl->minBits != 0ull || ctl->total != 0;
        __syncthreads();
        if (notMine) {
            if (tid == 0) p.nf[b] = -2;
            signal_done();
            return;
        }
        if (tid == 0) ctl->minBits = 0x7ff0000000000000ull;
    }
    __syncthreads();
    const int nLc = N - M;  // condL (assignment.cpp:60)
    // Threads take prefixes of the first M-2 columns and loop over the last two -- or, when those prefixes are too few to
    // occupy the workgroup, prefixes of M-1 columns and loop over the last one.
    u32 nPre2 = 1u;
    for (int c = 0; c < M - 2; c++) nPre2 *= (u32)(N - c);
    const bool inner2 = nPre2 >= (u32)(NT / 4);
    const u32 nPre = inner2 ? nPre2 : nPre2 * (u32)(N - (M - 2));
    const double scale = (double)TN_BUCKETS / (TN_GATE * (double)M);
    // every assignment that exists (feasible or not): few enough for the list -> one pass collects them all
    const bool all = (unsigned long long)nPre2 * (unsigned)(N - (M - 2)) * (unsigned)(N - (M - 1)) <= (unsigned long long)TINY_CAP;
    // A bound on everything that can be emitted, before any pass: the greedy assignment (column by column its cheapest free row) is
    // some assignment, so gainBest[0] <= its gain and nothing beyond greedy + cutoff is ever counted (cpp:709-719).
    if (wave == 0) {
        u64 used = 0ull;
        double gsum = 0.0;
        for (int c = 0; c < M; c++) {
            const double x = (lane < N && !((used >> lane) & 1ull)) ? Cs[lane + c * TN_LDT] : INF;
            const double m = wave_min_f64(x);
            const u64 at = __ballot(x == m && x < INF);
            if (!at) { gsum = INF; break; }
            const int rr = __builtin_ctzll(at);
            used |= 1ull << rr;
            gsum = gsum + m;
            if (lane == 0) gRow[c] = (unsigned char)rr;
        }
        if (lane == 0) {
            ctl->limit = (gsum < INF) ? gsum + p.cutoff : INF;
            ctl->greedyUsed = (gsum < INF) ? used : 0ull;
        }
    }
    __syncthreads();
    // A tighter bound where the frame is large: the greedy assignment's NEIGHBOURS -- one column on another free row, or two
    // columns on two other free rows -- are real assignments too; when at least k of them are feasible, the k-th smallest of
    // their gains (exact sums, as for any assignment) is an upper bound of the k-th best gain of all.  On dense frames this
    // is what keeps the passes from visiting every assignment below greedy + 42: the prefixes beyond it are dropped.
    {
        const u64 usedG = ctl->greedyUsed;
        const int F = N - M;  // free rows beside the greedy assignment
        const long long nSingle = (long long)M * F, nPair = (long long)(M * (M - 1) / 2) * F * (F - 1);
        // (not on frames whose passes are short anyway: the bound costs a pass over the neighbours and three barriers)
        const bool large = (unsigned long long)nPre2 * (unsigned)(N - (M - 2)) * (unsigned)(N - (M - 1)) > (1ull << 17);
        const bool tighten = usedG != 0ull && large && nSingle + nPair >= k && nSingle + nPair <= (1 << 16);
        if (tighten) {
            const u64 freeG = ((N >= 64) ? ~0ull : ((1ull << N) - 1ull)) & ~usedG;
            auto nth_free = [&](int f) {
                u64 m = freeG;
                for (int i = 0; i < f; i++) m &= m - 1ull;
                return __builtin_ctzll(m);
            };
            int cnt = 0;
            for (int t = tid; t < (int)(nSingle + nPair); t += NT) {
                int c1, c2 = -1, r1, r2 = -1;
                if (t < nSingle) {
                    c1 = t / F;
                    r1 = nth_free(t - c1 * F);
                } else {
                    int u = t - (int)nSingle;
                    const int per = F * (F - 1);
                    int pr = u / per;
                    u -= pr * per;
                    c1 = 0;
                    while (pr >= M - 1 - c1) { pr -= M - 1 - c1; c1++; }  // pair number -> (c1 < c2)
                    c2 = c1 + 1 + pr;
                    const int f1 = u / (F - 1);
                    int f2 = u - f1 * (F - 1);
                    f2 += (f2 >= f1) ? 1 : 0;
                    r1 = nth_free(f1);
                    r2 = nth_free(f2);
                }
                double g = 0.0;
                for (int c = 0; c < M; c++) {  // calcGain's order
                    const int r = (c == c1) ? r1 : (c == c2) ? r2 : (int)gRow[c];
                    g = g + Cs[r + c * TN_LDT];
                }
                if (g < INF) {
                    int bk = (int)(g * scale);
                    bk = bk > TN_BUCKETS - 1 ? TN_BUCKETS - 1 : bk;
                    atomicAdd(&hist[bk], 1u);
                    cnt++;
                }
            }
            if (cnt) atomicAdd(&ctl->total, cnt);
        }
        __syncthreads();
        if (tighten && ctl->total >= k && wave == 0) {  // the bucket of the k-th smallest neighbour: its upper edge bounds the k-th best
            constexpr int PER = TN_BUCKETS / 64;
            u32 mine = 0;
            for (int i = 0; i < PER; i++) mine += hist[lane * PER + i];
            u32 incl = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const u32 t = (u32)__shfl_up((int)incl, d);
                if (lane >= d) incl += t;
            }
            const u32 excl = incl - mine;
            if ((int)excl < k && (int)incl >= k) {
                u32 run = excl;
                for (int i = 0; i < PER; i++) {
                    run += hist[lane * PER + i];
                    if ((int)run >= k) {
                        const int b0 = lane * PER + i;
                        const double edge = (double)(b0 + 1) / scale * (1.0 + 1e-12);
                        if (b0 < TN_BUCKETS - 1 && edge < ctl->limit) ctl->limit = edge;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (tighten) {
            for (int i = tid; i < TN_BUCKETS; i += NT) hist[i] = 0u;
            if (tid == 0) ctl->total = 0;
        }
        __syncthreads();
    }
    const double limit = ctl->limit;
    // ---- the feasible prefixes, once: decoded (divisions, free-row walks), checked against +inf and the bound, kept in a list;
    //      the passes below then share the list's work items evenly.  More of them than the list holds: the plain walk.
    for (u32 q = (u32)tid; q < nPre; q += NT) {
        u64 rows, used;
        double acc;
        if (!tiny_prefix(Cs, N, inner2 ? M - 2 : M - 1, q, rows, used, acc) || acc > limit) continue;
        const int pos = atomicAdd(&ctl->nFeas, 1);
        if (pos < PCAP) {
            preRows[pos] = rows;
            preAcc[pos] = acc;
            preQ[pos] = q;
        }
    }
    __syncthreads();
    const int nFeas = ctl->nFeas;
    const bool listed = nFeas <= PCAP;
    // ---- pass 1: minimum and histogram (or, when everything fits the list, the list itself) -----------------------------------
    {
        double mn = INF;
        int cnt = 0;
        auto visit = [&](double g, u32 id) {
            mn = min_keep(mn, g);
            cnt++;
            if (all) {
                const int pos = atomicAdd(&ctl->listN, 1);
                listG[pos] = g;
                listI[pos] = id;
            } else {
                int bk = (int)(g * scale);
                bk = bk > TN_BUCKETS - 1 ? TN_BUCKETS - 1 : bk;
                atomicAdd(&hist[bk], 1u);
            }
        };
        if (listed) {
            if (inner2) tiny_walk_list<2>(Cs, N, M, preRows, preAcc, preQ, nFeas, tid, NT, limit, visit);
            else tiny_walk_list<1>(Cs, N, M, preRows, preAcc, preQ, nFeas, tid, NT, limit, visit);
        } else {
            if (inner2) tiny_walk<2>(Cs, N, M, nPre, tid, NT, limit, visit);
            else tiny_walk<1>(Cs, N, M, nPre, tid, NT, limit, visit);
        }
        mn = wave_min_f64(mn);
        if (lane == 0 && mn < INF) atomicMin(&ctl->minBits, (unsigned long long)__double_as_longlong(mn));
        if (cnt) atomicAdd(&ctl->total, cnt);
    }
    __syncthreads();
    const int total = ctl->total;
    if (total == 0) {  // infeasible: kBest2D returns 0 (cpp:588-593)
        if (tid == 0) p.nf[b] = 0;
        signal_done();
        return;
    }
    const double best = __longlong_as_double((long long)ctl->minBits);  // gainBest[0] (CDelta = 0 on a conditioned matrix)
    const double cutG = best + p.cutoff;                                  // cpp:681
    if (!all) {
        // the first bucket at which the cumulated count reaches k
        if (wave == 0) {
            constexpr int PER = TN_BUCKETS / 64;
            u32 mine = 0;
            for (int i = 0; i < PER; i++) mine += hist[lane * PER + i];
            u32 incl = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const u32 t = (u32)__shfl_up((int)incl, d);
                if (lane >= d) incl += t;
            }
            const u32 excl = incl - mine;
            if ((int)excl < k && (int)incl >= k) {  // exactly one lane (when total >= k)
                u32 run = excl;
                for (int i = 0; i < PER; i++) {
                    run += hist[lane * PER + i];
                    if ((int)run >= k) { ctl->bStar = lane * PER + i; break; }
                }
            }
        }
        __syncthreads();
        const int bStar = ctl->bStar;
        // ---- pass 2: collect the assignments up to that bucket -------------------------------------------------------------
        auto collect = [&](double g, u32 id) {
            int bk = (int)(g * scale);
            bk = bk > TN_BUCKETS - 1 ? TN_BUCKETS - 1 : bk;
            if (bk <= bStar && !(g > cutG)) {  // (what lies beyond the cutoff is never emitted: cpp:709-719)
                const int pos = atomicAdd(&ctl->listN, 1);
                if (pos < TINY_CAP) {
                    listG[pos] = g;
                    listI[pos] = id;
                }
            }
        };
        // (nothing above the k-th bucket's upper edge, nothing beyond the cutoff)
        const double edge = (double)(bStar + 1) / scale * (1.0 + 1e-12);
        const double lim2 = (bStar < TN_BUCKETS - 1 && edge < cutG) ? edge : (cutG < limit ? cutG : limit);
        if (listed) {  // (the list was made against the looser bound: a prefix beyond lim2 gives its items nothing to do)
            if (inner2) tiny_walk_list<2>(Cs, N, M, preRows, preAcc, preQ, nFeas, tid, NT, lim2, collect);
            else tiny_walk_list<1>(Cs, N, M, preRows, preAcc, preQ, nFeas, tid, NT, lim2, collect);
        } else {
            if (inner2) tiny_walk<2>(Cs, N, M, nPre, tid, NT, lim2, collect);
            else tiny_walk<1>(Cs, N, M, nPre, tid, NT, lim2, collect);
        }
    }
    for (int i = tid; i < TINY_CAP; i += NT) rankA[i] = 0;
    __syncthreads();
    const int n = ctl->listN;
    if (n > TINY_CAP) {  // (thousands of gains in one bucket: the enumeration kernels take the frame)
        if (tid == 0) p.nf[b] = -2;
        signal_done();
        return;
    }
    // ---- rank sort by (gain, index); several threads share an element's comparisons -----------------------------------------
    {
        const int per = (n > 0 && NT / n > 0) ? NT / n : 1;  // threads per element
        for (int e0 = 0; e0 < n; e0 += NT / per) {
            const int e = e0 + tid / per, part = tid % per;
            if (e < n && tid / per < NT / per) {
                const double g = listG[e];
                const u32 id = listI[e];
                int rk = 0, j = part;
                for (; j + 3 * per < n; j += 4 * per) {  // (four reads in flight)
                    double g2[4];
                    u32 i2[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        g2[q] = listG[j + q * per];
                        i2[q] = listI[j + q * per];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) rk += (g2[q] < g || (g2[q] == g && i2[q] < id)) ? 1 : 0;
                }
                for (; j < n; j += per) {
                    const double g2 = listG[j];
                    rk += (g2 < g || (g2 == g && listI[j] < id)) ? 1 : 0;
                }
                if (rk) atomicAdd(&rankA[e], rk);
            }
        }
    }
    for (int e = tid; e < n; e += NT)  // cpp:709-719: solutions are counted while not beyond gainBest[0] + cutoff
        if (!(listG[e] > cutG)) atomicAdd(&ctl->nWithin, 1);
    __syncthreads();
    int nf = ctl->nWithin;  // (those within the cutoff are a prefix of the ascending order)
    nf = nf < k ? nf : k;
    for (int e = tid; e < n; e += NT) {
        const int rk = rankA[e];
        if (rk < nf) {
            solG[rk] = listG[e];
            solI[rk] = listI[e];
        }
    }
    __syncthreads();
    // exact ties (kbest_ties.h): the list holds every assignment up to the k-th gain's bucket and is ordered by (gain, id) -- id
    // grows with the rows of the columns 0, 1, ...: the canonical order -- so the k kept are the lexicographically first of their
    // gain level whatever its size; the flag says that there WAS a choice (the solution of rank k has the k-th gain)
    if (p.tieFlags && nf == k)
        for (int e = tid; e < n; e += NT)
            if (rankA[e] == k && listG[e] == solG[k - 1] && !(listG[e] > cutG)) p.tieFlags[b] = KBEST_TIE_BOUNDARY | KBEST_TIE_RESOLVED;
    // ---- the solutions' rows, the weights (assignment.cpp:616-648), as in kbest_small.hip ----------------------------------
    for (int s = tid; s < nf; s += NT) {
        const u32 id = solI[s];
        u64 rows, used;
        double acc;
        const int D = inner2 ? M - 2 : M - 1;
        tiny_prefix(Cs, N, D, inner2 ? id >> 12 : id >> 6, rows, used, acc);
        // (every unassigned measurement's own row counts as "no landmark": row nLc, :634-637)
        for (int c = 0; c < D; c++) {
            const int r = (int)((rows >> (8 * c)) & 0xffull);
            rTab[s * M + c] = (unsigned char)(r >= nLc ? nLc : r);
        }
        if (inner2) {
            const int r = (int)((id >> 6) & 63u);
            rTab[s * M + M - 2] = (unsigned char)(r >= nLc ? nLc : r);
        }
        {
            const int r = (int)(id & 63u);
            rTab[s * M + M - 1] = (unsigned char)(r >= nLc ? nLc : r);
        }
        const double g = solG[s];
        wts[s] = (p.gate && !(best + TN_GATE > g)) ? 0.0 : exp(best - g);  // :622-626 (0.0: skipped -- x + 0.0 is x, bit for bit, for the x >= +0.0 here)
    }
    __syncthreads();
    const int nAcc = M * (nLc + 1);
    for (int i = tid; i < nAcc; i += NT) {
        const int accC = i / (nLc + 1), accR = i - accC * (nLc + 1);
        double total2 = 0.0, acc = 0.0;
        const unsigned char *rp = rTab + accC;
        // solutions ascending; total and every probs[col][row] summed sequentially (:633-638); eight solutions' reads in flight
        int s = 0;
        for (; s + 8 <= nf; s += 8) {
            double w[8];
            int r[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                w[q] = wts[s + q];
                r[q] = rp[(s + q) * M];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double a2 = acc + w[q];
                total2 = total2 + w[q];
                acc = (r[q] == accR) ? a2 : acc;
            }
        }
        for (; s < nf; s++) {
            const double w = wts[s];
            const double a2 = acc + w;
            total2 = total2 + w;
            acc = ((int)rp[s * M] == accR) ? a2 : acc;
        }
        const double norm = 1.0 / total2;  // :643
        // scatter back to the caller's landmark numbering (getAssignmentProbs, assignment.cpp:68-74)
        const int ro = (accR >= nLc) ? nLout : (int)rowIdx[accR];
        probOut[accC * (nLout + 1) + ro] = acc * norm;
    }
    if (tid == 0) p.nf[b] = nf;
    signal_done();
}

int tiny_lds_bytes(int k, int nThreads)
{
    int o = TINY_MAX_COL * TN_LDT * 8 + TINY_MAX_COL * TINY_MAX_ROW * 8 + TINY_MAX_COL * 8 + 56 + 8 + TINY_MAX_ROW * 2 +
            TN_BUCKETS * 4 + TINY_CAP * 16 + k * 12;
    o = (o + 7) & ~7;
    o += k * 8 + k * TINY_MAX_COL;
    o = (o + 7) & ~7;
    return o + tiny_prefix_cap(nThreads) * 20 + 16;
}

template <int NT>
static hipError_t launch_tiny_nt(const SmallParams &p, int B, hipStream_t stream)
{
    const int lds = tiny_lds_bytes(p.kTab, NT);
    static std::atomic<int> granted[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > granted[dev & 15].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_tiny_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        granted[dev & 15].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((kbest_tiny_kernel<NT>), dim3(B), dim3(NT), lds, stream, p);
    return hipGetLastError();
}

// many = the batch fills the chip: smaller workgroups
hipError_t launch_kbest_tiny(const SmallParams &p, int B, bool many, hipStream_t stream)
{
    return many ? launch_tiny_nt<256>(p, B, stream) : launch_tiny_nt<1024>(p, B, stream);
}

}  // namespace kb
