// kbest_bnb.hip -- MI355X (gfx950): the fused association path (cost block in -> probabilities out) by a BOUNDED WALK.
//
// What getAssignmentProbs needs of kBest2DCutoff (assignment.cpp:57-74, 594; shortestPathCPP.cpp:646-733) is the k cheapest
// assignments of the conditioned problem in ascending order.  Murty's enumeration finds them one shortest-path search at a
// time -- on this device rounds of ~15 us, ten to thirty of them per frame.  But conditioned entries are >= 0
// (conditionCosts subtracts the column minima, assignment.cpp:476-496): the partial sums of an assignment, added column by
// column in calcGain's order (shortestPathCPP.cpp:59-80), only grow, so a walk over the columns (level by level) that drops a
// branch as soon as its partial sum exceeds a bound U visits EVERY assignment with gain <= U and little else.  With U
// raised until k assignments lie below it, the k best are the k smallest of what the walk collected -- their gains are
// the very sums calcGain computes (same additions, same order: same bits), and nothing about them depends on how they
// were found.  On a 28 x 10 KITTI-like frame with k = 200 the bound search visits ~7 000 partial assignments in all its
// passes (host model tests/dev/proto_bnb.py) where the enumeration needs ~400 shortest-path searches of 28 rows.
//
//   * the bound: a partial assignment is dropped when its partial sum PLUS a lower bound of what the columns still to come
//     must add exceeds U -- per column its cheapest entry, or its second cheapest where the partial assignment has used the
//     cheapest one's row (the rows of a column are sorted by cost once).  With it a level holds about as many partial
//     assignments as there are assignments below U; without it frames whose cheapest rows collide overflow any list just above
//     their best gain;
//   * one pass: breadth first, level by level, the columns with the fewest feasible rows first; the partial assignments of a
//     level lie in one LDS list (rows packed a byte per column, the set of used rows, the partial sum), their children are
//     written to the other; a (partial assignment, feasible row) pair is one thread's work; one barrier per level.  An
//     assignment that reaches the last level has its gain summed again in calcGain's order, from 0.0, left to right;
//   * the bound search: U starts at the gain of the greedy assignment (column by column its cheapest free row; a real
//     assignment, so gainBest[0] <= it), every pass fills a 512-bucket histogram over [0, U]; a pass that counts fewer than
//     k assignments raises U (by the growth exponent of the last two counts, aimed at 1.5 k).  A pass whose level outgrows
//     its list does not start over: it histograms the lower bounds of that level's children over (the last bound that fitted,
//     U], lowers U to the largest bucket edge at which they fit, and goes on under that bound -- what it counts in the end is
//     exact for the bound it returns (levels walked under the looser bound only carried entries that die now).  A lowered pass
//     that still counts fewer than k narrows the window 512-fold per pass; when it stops moving no bound both fits and has k
//     below it.  Nothing beyond greedy + cutoff is ever emitted (cpp:705-719), so U
//     stops there.  The first pass with k assignments below its bound also gives the bucket of the k-th gain; one more walk
//     bounded by that bucket's upper edge collects the candidates (k + ~4), a rank sort by (gain, rows) orders them, the
//     first min(k, those within the cutoff) are the solutions; then the weights exactly as in kbest_small.hip;
//   * a frame for which no bound has k assignments below it AND levels that fit (masses of equal or nearly equal gains)
//     comes back with nf = -2 and is answered by the enumeration kernel -- re-run alone by the host entry, by a second launch
//     that looks at nothing else behind the device entry.  The order of exactly equal gains is by rows here, by the
//     enumeration tree there and by the heap in the reference: only when different assignments with
//     equal gains straddle slot k can the emitted sets differ.
// fp64 add / compare / exp only; no fast-math.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "kbest_engine.h"
#include "kbest_wave.h"

namespace kb {

namespace {

constexpr double BN_GATE = 42.0;  // assignment.cpp:9
constexpr int BN_BUCKETS = 512;
#ifndef KB_BNB_FILL
#define KB_BNB_FILL 16  // sixteenths of a list that a lowered bound may fill
#endif
#ifndef KB_BNB_AIM
#define KB_BNB_AIM 1.5   // a bound that counted fewer than k is raised to where this many k are expected,
#define KB_BNB_FMAX 2.0  // by this factor at most
#endif
#ifndef KB_BNB_FMAX_LONE
#define KB_BNB_FMAX_LONE 3.0  // ... for a frame by itself (1 024 threads, bisection instead of lowering: one frame per call 78 -> 75 us; 4.0: the same)
#endif
#ifndef KB_BNB_FILLK
#define KB_BNB_FILLK 1000.0  // ... and no more than this many k (no gain from limiting it: measured 2.5, 4)
#endif
#ifndef KB_BNB_LOWER
#define KB_BNB_LOWER 2  // in-pass lowering of the bound: 0 never, 1 always, 2 in the 256-thread (batch) shape only -- a batch lasts as
#endif                  // long as its slowest frame (lowering cuts the passes of those: C5 0.32 -> 0.23 ms), a frame by itself is
                        // ~6 % slower with it (the lowered bound fills the 1 536-entry lists: more work than a bisection step)
#ifndef KB_BNB_START
#define KB_BNB_START 1.0  // the first bound, in greedy gains
#endif
#ifndef KB_BNB_KEEP
#define KB_BNB_KEEP 1  // 1: the counting passes keep what they count (gain, rows) while it fits the candidate list; the pass that finds k
#endif                 // below its bound then only drops what lies beyond the k-th gain's bucket; where the list did not hold them all,
                       // the LAST level alone is walked again over the frontier that pass left -- no collecting walk either way
                       // (round 6, 1 000 frames: 0.213 -> 0.191 ms, one frame per call 90 -> 80 us)
#ifndef KB_BNB_CH
#define KB_BNB_CH 1    // 1: a counting pass that may lower its bound histograms its children's lower bounds AS IT WALKS a level (the
#endif                 // two halves of the leaf histogram, level by level): an overflowing level is walked twice, not three times
                       // (round 6: 0.191 -> 0.165 ms; the launch lasts as long as its slowest frame, and those are the lowered ones)
#ifndef KB_BNB_AGG
#define KB_BNB_AGG 0   // 1: a wave appends its children with ONE LDS atomic (the lanes that reach the append in the same step):
#endif                 // measured 2 % SLOWER (0.213 -> 0.217 ms) -- the per-child atomics are not what a level costs

struct BEntry {  // a partial assignment: rows of the columns 0 .. level-1 (one byte each), the set of those rows, their sum
    u64 rowsLo, rowsHi, used;
    double acc;
};

struct BCtrl {
    unsigned long long minBits;  // smallest gain seen (non-negative doubles order like their bits)
    int nA, nB;                  // entries of the two frontier lists
    int count;                   // assignments below the bound in this pass
    int listN;                   // candidates collected
    int nWithin;                 // ... of which within the cutoff
    int next;                    // depth-first phase: next frontier entry to take
    int nodes;                   // partial assignments visited in this pass
    int abort;                   // the pass ran over its budget
    int bStar;                   // bucket of the k-th smallest gain
    int negative;                // (unconditioned input) a negative entry: not this kernel's problem
    double limit;                // greedy + cutoff
    double maxFinite;            // largest finite entry
};
static_assert(sizeof(BCtrl) <= 80, "the LDS carve-up of kbest_bnb_kernel gives BCtrl 80 bytes");

__device__ __forceinline__ int row_of(u64 lo, u64 hi, int c) { return (int)(((c < 8 ? lo : hi) >> (8 * (c & 7))) & 0xffull); }
__device__ __forceinline__ void set_row(u64 &lo, u64 &hi, int c, int r)
{
    const u64 m = 0xffull << (8 * (c & 7)), v = (u64)r << (8 * (c & 7));
    if (c < 8) lo = (lo & ~m) | v;
    else hi = (hi & ~m) | v;
}

// the next `n = 1` slots of an LDS counter for every lane that is active here: one atomic per wave (KB_BNB_AGG) or one per lane
__device__ __forceinline__ int bn_append(int *counter)
{
#if KB_BNB_AGG
    const u64 act = __ballot(1);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __builtin_ctzll(act)) base = atomicAdd(counter, __popcll(act));
    base = __builtin_amdgcn_readfirstlane(base);
    return base + __popcll(act & ((1ull << lane) - 1ull));
#else
    return atomicAdd(counter, 1);
#endif
}

}  // namespace

template <int NT>
__global__ void __launch_bounds__(NT) kbest_bnb_kernel(SmallParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWV = NT / 64;
    constexpr int FCAP = bnb_frontier_cap(NT);
    const double INF = d_inf();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int k = p.kTab;  // (the caller's k; p.k is what the ENUMERATION kernels enumerate: one more, kbest_ties.h)
    if (p.tieFlags && threadIdx.x == 0) p.tieFlags[blockIdx.x] = 0;
    const int M = p.imm ? p.immCol : p.nCol[b];
    const int NR = p.imm ? p.immRow : p.nRow[b];
    const int nLout = p.imm ? p.immL : p.nL[b];
    const unsigned long long dStart = __builtin_readcyclecounter();
    const double *Cg = p.cost + (p.costOff ? p.costOff[b] : 0);
    double *probOut = p.probs + (p.probOff ? p.probOff[b] : 0);
    // LDS: tile | column minima | control | kept rows | row index | histogram | frontier lists A, B (the raw block first lies in A) |
    //      candidates (gain, rows) | rank | solutions | weights | row table
    // (tile, row lists: for the largest block of the LAUNCH -- what is not needed is room for one more workgroup on the CU)
    const int tileCol = p.maxCol < BNB_MAX_COL ? p.maxCol : BNB_MAX_COL;
    const int tileRow = (p.bnbRow > 0 && p.bnbRow < BNB_MAX_ROW) ? p.bnbRow : BNB_MAX_ROW;
    const int BN_LDT = tileRow + 1;
    int o = 0;
    double *Cs = reinterpret_cast<double *>(smem + o);       o += tileCol * BN_LDT * 8;
    double *colMin = reinterpret_cast<double *>(smem + o);   o += BNB_MAX_COL * 8;
    BCtrl *ctl = reinterpret_cast<BCtrl *>(smem + o);        o += 80;
    u64 *keepW = reinterpret_cast<u64 *>(smem + o);          o += 8;
    unsigned short *rowIdx = reinterpret_cast<unsigned short *>(smem + o);  o += BNB_MAX_ROW * 2;
    u32 *hist = reinterpret_cast<u32 *>(smem + o);           o += BN_BUCKETS * 4;
    unsigned char *feasRow = smem + o;                       o += (tileCol * tileRow + 7) & ~7;  // per column: its rows with a finite entry
    unsigned char *nFeasRow = smem + o;                      o += BNB_MAX_COL;
    int *lvlN = reinterpret_cast<int *>(smem + o);           o += (BNB_MAX_COL + 1) * 4 + 12;  // frontier entries per level
    unsigned char *ord = smem + o;                           o += BNB_MAX_COL;                 // the walk's column order
    u64 *zMask = reinterpret_cast<u64 *>(smem + o);          o += BNB_MAX_COL * 8;
    u64 *zBit = reinterpret_cast<u64 *>(smem + o);           o += BNB_MAX_COL * 8;
    double *c0 = reinterpret_cast<double *>(smem + o);       o += BNB_MAX_COL * 8;
    double *c1 = reinterpret_cast<double *>(smem + o);       o += BNB_MAX_COL * 8;
    double *c0Sum = reinterpret_cast<double *>(smem + o);    o += BNB_MAX_COL * 8;
    BEntry *listA = reinterpret_cast<BEntry *>(smem + o);    o += FCAP * 32;
    BEntry *listB = reinterpret_cast<BEntry *>(smem + o);    o += FCAP * 32;
    double *stage = reinterpret_cast<double *>(listA);       // (BNB_MAX_ROW x BNB_MAX_COL doubles = 8 KB <= the two lists)
    // The candidates of the last walk lie in the frontier list that its last level does not read (the levels alternate between
    // the two lists); the solutions, their weights and rows then take the other one, dead by then.
    constexpr int CAP = bnb_cand_cap(NT);
    unsigned char *lastB = reinterpret_cast<unsigned char *>(((M - 1) & 1) ? listA : listB);
    unsigned char *lastA = reinterpret_cast<unsigned char *>(((M - 1) & 1) ? listB : listA);
    double *candG = reinterpret_cast<double *>(lastB);
    u64 *candLo = reinterpret_cast<u64 *>(lastB + CAP * 8);
    u64 *candHi = reinterpret_cast<u64 *>(lastB + CAP * 16);
    int *rankA = reinterpret_cast<int *>(lastB + CAP * 24);
    double *solG = reinterpret_cast<double *>(lastA);
    double *wts = reinterpret_cast<double *>(lastA + k * 8);
    unsigned char *rTab = lastA + k * 16;                    // [k][16]

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
    if (M < 1 || M > tileCol || NR > tileRow || NR < M || nLout + M != NR) {  // not what this kernel takes
        if (tid == 0) p.nf[b] = -2;
        signal_done();
        return;
    }
    for (int i = tid; i < M * (nLout + 1); i += NT) probOut[i] = 0.0;
    for (int i = tid; i < NR * M; i += NT) stage[i] = Cg[i];
    for (int i = tid; i < tileCol * BN_LDT; i += NT) Cs[i] = INF;
    if (tid == 0) {
        ctl->minBits = 0x7ff0000000000000ull;
        ctl->listN = 0;
        ctl->nWithin = 0;
        ctl->negative = 0;
        ctl->limit = INF;
        ctl->maxFinite = 0.0;
    }
    __syncthreads();
    const unsigned long long dSA = __builtin_readcyclecounter();
    int N;
    if (p.condition) {
        // ---- conditionCosts (assignment.cpp:439-525), as in kbest_small.hip ----------------------------------------------
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
                for (int c = 0; c < M; c++) good = good | (stage[c * NR + lane] <= colMin[c] + BN_GATE);
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
                Cs[nr + c * BN_LDT] = (x <= colMin[c] + BN_GATE) ? (x - colMin[c]) : INF;
            }
        }
    } else {
        // assignmentProb on a block that is conditioned already (assignment.cpp:58-62).  kBest2DCutoff shifts the matrix by its
        // smallest entry (makeCostMatrixSafe, cpp:534-569) and adds CDelta * numCol back to every gain (cpp:583, 626-630): with an
        // exact 0.0 as the smallest entry -- every column of a conditioned block holds one -- the shift is the identity and the
        // gains are calcGain's sums as they stand.  Any other block is not this kernel's: -2.
        N = NR;
        double mn = INF;
        for (int i = tid; i < NR * M; i += NT) {
            const double x = stage[i];
            mn = min_keep(mn, x);
            const int c = i / NR, r = i - c * NR;
            Cs[r + c * BN_LDT] = (x == x) ? x : INF;
        }
        mn = wave_min_f64(mn);
        if (lane == 0) {
            if (mn < 0.0) atomicAdd(&ctl->negative, 1);
            else atomicMin(&ctl->minBits, (unsigned long long)__double_as_longlong(mn));
        }
        if (tid < NR) rowIdx[tid] = (unsigned short)tid;
        __syncthreads();
        const bool notMine = ctl->minBits != 0ull || ctl->negative != 0;
        __syncthreads();
        if (notMine) {
            if (tid == 0) p.nf[b] = -2;
            signal_done();
            return;
        }
        if (tid == 0) ctl->minBits = 0x7ff0000000000000ull;
    }
    __syncthreads();
    const unsigned long long dSB = __builtin_readcyclecounter();
    const int nLc = N - M;  // condL (assignment.cpp:60)
    // ---- single column: assignmentProb's fast path (assignment.cpp:554-570), as in kbest_small.hip -------------------------
    if (M == 1) {
        if (tid == 0) {
            double norm = 0.0;
            int cnt = 0;
            for (int i = 0; i <= nLc; i++) {
                const double c = Cs[i];
                if (c < BN_GATE) { norm += exp(-c); cnt++; }
            }
            norm = 1.0 / norm;
            for (int i = 0; i <= nLc; i++) {
                const double c = Cs[i];
                const double q = (c < BN_GATE) ? exp(-c) : 0.0;
                probOut[(i >= nLc) ? nLout : (int)rowIdx[i]] = q * norm;
            }
            p.nf[b] = cnt < k ? cnt : k;
        }
        signal_done();
        return;
    }
    // the largest finite entry (an upper bound of any gain: M times it), and the greedy assignment's gain
    {
        double mx = 0.0;
        for (int i = tid; i < M * N; i += NT) {
            const int c = i / N, r = i - c * N;
            const double x = Cs[r + c * BN_LDT];
            if (x < INF && x > mx) mx = x;
        }
        for (int d = 32; d; d >>= 1) {
            const double t = __shfl_xor(mx, d);
            mx = t > mx ? t : mx;
        }
        if (lane == 0 && mx > 0.0) atomicMax(reinterpret_cast<unsigned long long *>(&ctl->maxFinite), (unsigned long long)__double_as_longlong(mx));
    }
    double gsum = INF;
    if (wave == 0) {
        u64 used = 0ull;
        double gs = 0.0;
        for (int c = 0; c < M; c++) {
            const double x = (lane < N && !((used >> lane) & 1ull)) ? Cs[lane + c * BN_LDT] : INF;
            const double m = wave_min_f64(x);
            const u64 at = __ballot(x == m && x < INF);
            if (!at) { gs = INF; break; }
            used |= 1ull << __builtin_ctzll(at);
            gs = gs + m;
        }
        if (lane == 0) ctl->limit = gs;
    }
    __syncthreads();
    const unsigned long long dSC = __builtin_readcyclecounter();
    gsum = ctl->limit;
    const double gAll = ctl->maxFinite * (double)M * (1.0 + 1e-12) + 1e-300;  // no gain lies beyond this
    // nothing beyond greedy + cutoff is ever emitted (cpp:705-719: gainBest[0] <= greedy)
    const double Umax = (gsum < INF && gsum + p.cutoff < gAll) ? gsum + p.cutoff : gAll;

    // the rows of each column that can be taken at all (finite entries), cheapest first: a frontier entry is only tried against
    // those, in that order (the first one beyond the bound ends the column), and the cheapest row of a column that an entry has
    // not used yet is a lower bound of what that column will add
    for (int c = wave; c < M; c += NWV) {
        const double mine = lane < N ? Cs[lane + c * BN_LDT] : INF;
        const bool fin = mine < INF;
        int rank = 0;
        for (int j = 0; j < N; j++) {
            const double other = Cs[j + c * BN_LDT];
            rank += (other < INF && (other < mine || (other == mine && j < lane))) ? 1 : 0;
        }
        if (fin) feasRow[c * tileRow + rank] = (unsigned char)lane;
        const u64 m = __ballot(fin);
        if (lane == 0) nFeasRow[c] = (unsigned char)__popcll(m);
    }
    __syncthreads();
    const unsigned long long dSD = __builtin_readcyclecounter();
    // The walk takes the columns with the fewest feasible rows first (fewer partial assignments per level: 208 instead of 330 at
    // most on the KITTI-like frames, host model); its partial sums -- in walk order -- only prune, with a margin far above their
    // rounding; the gain of an assignment that reaches the last level is summed again in calcGain's order.
    if (wave == 0) {
        const int nfl = lane < M ? (int)nFeasRow[lane] : 1 << 20;
        int rank = 0;
        for (int j = 0; j < M; j++) {
            const int nj = __builtin_amdgcn_readlane(nfl, j);
            rank += (nj < nfl || (nj == nfl && j < lane)) ? 1 : 0;
        }
        if (lane < M) ord[rank] = (unsigned char)lane;
        wave_fence();
        // Lower bound of what the columns behind a level must still add: each its cheapest entry -- or, where an entry has
        // used that row, its second cheapest (which bounds every other row of the column from below).  By walk level: zBit the
        // cheapest row's bit, c0 / c1 the two costs; zMask[level] the cheapest rows of all columns behind `level` (an entry that
        // has used none of them is bounded by the sum of the c0 alone, c0Sum[level])
        if (lane < M) {
            const int c = ord[lane], n = nFeasRow[c];
            const unsigned char *f = feasRow + c * tileRow;
            zBit[lane] = n >= 1 ? 1ull << f[0] : 0ull;
            c0[lane] = n >= 1 ? Cs[f[0] + c * BN_LDT] : INF;
            c1[lane] = n >= 2 ? Cs[f[1] + c * BN_LDT] : INF;
        }
        wave_fence();
        if (lane < M) {
            u64 z = 0ull;
            double s0 = 0.0;
            for (int l2 = lane + 1; l2 < M; l2++) { z |= zBit[l2]; s0 = s0 + c0[l2]; }
            zMask[lane] = z;
            c0Sum[lane] = s0;
        }
    }
    __syncthreads();

    // ---- one walk: every assignment whose partial sums stay <= U reaches `leaf`; breadth first, level by level: every
    //      (partial assignment, feasible row) pair is one thread's work.  A level whose children do not fit the list ends the
    //      walk (the bound is too large for this kernel's lists: the search below steps back).
    // mode 0: count + histogram over [0, U] + minimum; mode 1: collect (gain, rows) of those <= U
    // A counting pass (mode 0) whose frontier outgrows the lists at some level does not give up: it histograms the lower bounds
    // of that level's children, lowers the bound to the largest that still fits a list and goes on from there -- what it
    // counts in the end is exact for the bound it RETURNS.
    // from: 0, or M - 1 -- the LAST level alone, over the frontier the previous (complete) walk left in its list: what that walk
    // counted below its bound, collected below a tighter one, for the price of one level
    auto walk = [&](double U, const int mode, const double base, const int from = 0) -> double {  // (base: a bound known to fit the lists; < 0: no lowering)
        double scale = (double)BN_BUCKETS / (U > 0.0 ? U : 1.0);
        double Uprune = U * (1.0 + 1e-12);
        int tight = 0;
        double mn = INF;
        int cnt = 0;
        if (tid == 0) {
            if (from == 0) { listA[0].rowsLo = 0ull; listA[0].rowsHi = 0ull; listA[0].used = 0ull; listA[0].acc = 0.0; }
            ctl->count = 0;
            ctl->abort = 0;
            ctl->nodes = 0;
            if (KB_BNB_KEEP) ctl->listN = 0;
        }
        if (from == 0 && tid <= M) lvlN[tid] = tid == 0 ? 1 : 0;
        if (mode == 0)
            for (int i = tid; i < BN_BUCKETS; i += NT) hist[i] = 0u;
        __syncthreads();
        BEntry *A = (from & 1) ? listB : listA, *B = (from & 1) ? listA : listB;
        for (int level = from; level < M; level++) {
            const int nA = lvlN[level];
            if (nA == 0) break;
            const bool last = level == M - 1;
            const int col = ord[level];
            const double *Ccol = Cs + col * BN_LDT;
            const int nf_ = nFeasRow[col];
            const unsigned char *fr = feasRow + col * tileRow;
            const u64 zHere = zMask[level];
            const double c0Here = c0Sum[level];
            // 2^sh threads share a frontier entry's feasible rows (as many as keep the workgroup busy; no division)
            int sh = 0;
            while (sh < 5 && (nA << (sh + 1)) <= NT && (1 << sh) < nf_) sh++;
            bool stop = false;  // (alike in every thread)
#if !KB_BNB_CH
            bool histo = false;
#endif
#if KB_BNB_CH
            // the children's lower bounds, histogrammed over (base, U] while the level is walked: half (level & 1) of `hist` (the leaf
            // histogram: only the LAST level needs it), 256 buckets; the other half is cleared for the next level meanwhile
            constexpr int CHB = BN_BUCKETS / 2;
            const bool useCH = mode == 0 && base >= 0.0 && !last;
            u32 *hc = hist + (level & 1) * CHB;
#endif
            for (;;) {
#if KB_BNB_CH
            const double cscale = (double)CHB / (U - base);
            if (useCH)  // (dead since the barrier that ended level - 1; before the last level both halves are zero again)
                for (int i = tid; i < CHB; i += NT) hist[((level + 1) & 1) * CHB + i] = 0u;
#endif
            for (int e = tid >> sh; e < nA; e += NT >> sh)
            for (int j = tid & ((1 << sh) - 1); j < nf_; j += 1 << sh) {
                const int r = fr[j];
                const u64 used = A[e].used;
                double a = A[e].acc + Ccol[r];
                if (!(a <= Uprune)) break;  // (cheapest first: the rest of the column is beyond the bound too)
                if ((used >> r) & 1ull) continue;
                // what the columns still to come must add at least: each its cheapest row that is still free
                const u64 used2 = used | (1ull << r);
                double lb = a + c0Here;
                if (used2 & zHere) {  // (some column behind has lost its cheapest row to this entry)
                    lb = a;
                    for (int l2 = level + 1; l2 < M; l2++) lb = lb + ((used2 & zBit[l2]) ? c1[l2] : c0[l2]);
                }
                if (!(lb <= Uprune)) continue;
#if KB_BNB_CH
                if (useCH) {
                    int bk = lb <= base ? 0 : (int)((lb - base) * cscale);
                    bk = bk > CHB - 1 ? CHB - 1 : bk;
                    atomicAdd(&hc[bk], 1u);
                }
#else
                if (histo) {  // (buckets over (base, U]: what is within base is known to fit)
                    int bk = lb <= base ? 0 : (int)((lb - base) * ((double)BN_BUCKETS / (U - base)));
                    bk = bk > BN_BUCKETS - 1 ? BN_BUCKETS - 1 : bk;
                    atomicAdd(&hist[bk], 1u);
                    continue;
                }
#endif
                u64 lo = A[e].rowsLo, hi = A[e].rowsHi;
                set_row(lo, hi, col, r);
                if (last) {
                    a = 0.0;  // calcGain (cpp:59-80): column by column from 0.0, left to right
                    for (int c = 0; c < M; c++) a = a + Cs[row_of(lo, hi, c) + c * BN_LDT];
                    if (!(a <= U)) continue;
                    if (mode == 0) {
                        mn = min_keep(mn, a);
                        cnt++;
                        int bk = (int)(a * scale);
                        bk = bk > BN_BUCKETS - 1 ? BN_BUCKETS - 1 : bk;
                        atomicAdd(&hist[bk], 1u);
                    }
                    if (mode != 0 || KB_BNB_KEEP) {  // (the last level reads the OTHER list: this one is free for the candidates)
                        const int pos = bn_append(&ctl->listN);
                        if (pos < CAP) { candG[pos] = a; candLo[pos] = lo; candHi[pos] = hi; }
                    }
                } else {
                    const int pos = bn_append(&lvlN[level + 1]);
                    if (pos < FCAP) { B[pos].rowsLo = lo; B[pos].rowsHi = hi; B[pos].used = used | (1ull << r); B[pos].acc = a; }
                }
            }
            __syncthreads();
#if KB_BNB_CH
            if (!last && lvlN[level + 1] > FCAP) {
                if (!useCH || tight >= 8) { stop = true; break; }
                // the last bucket up to which the children still fit a list: its upper edge is the new bound
                if (wave == 0) {
                    constexpr int PER = CHB / 64;
                    u32 mine = 0;
                    for (int i = 0; i < PER; i++) mine += hc[lane * PER + i];
                    u32 incl = mine;
                    for (int d = 1; d < 64; d <<= 1) {
                        const u32 t = (u32)__shfl_up((int)incl, d);
                        if (lane >= d) incl += t;
                    }
                    u32 run = incl - mine;
                    const u32 fillTo = (u32)(FCAP * KB_BNB_FILL / 16) < (u32)(KB_BNB_FILLK * k) ? (u32)(FCAP * KB_BNB_FILL / 16) : (u32)(KB_BNB_FILLK * k);
                    int fit = 0;  // buckets of this lane's share that still fit
                    for (int i = 0; i < PER; i++) {
                        run += hc[lane * PER + i];
                        if (run <= fillTo) fit = i + 1;
                    }
                    const u64 full = __ballot(fit == PER);
                    const int firstShort = full == ~0ull ? 64 : __ffsll((long long)~full) - 1;
                    const int nb = firstShort * PER + __shfl(fit, firstShort < 64 ? firstShort : 63);
                    if (lane == 0) ctl->limit = firstShort == 64 ? U : base + (double)nb * ((U - base) / (double)CHB);
                }
                __syncthreads();
                const double nu = ctl->limit;
                for (int i = tid; i < CHB; i += NT) hc[i] = 0u;
                if (tid == 0) lvlN[level + 1] = 0;
                __syncthreads();
                if (!(nu > base) || !(nu < U)) { stop = true; break; }  // (one bucket alone overfills the list)
                U = nu;
                scale = (double)BN_BUCKETS / U;
                Uprune = U * (1.0 + 1e-12);
                tight++;
                continue;  // (this level again, under the new bound)
            }
            if (useCH && level == M - 2) {  // the last level counts into the whole histogram: this level's half goes back to zero
                for (int i = tid; i < CHB; i += NT) hc[i] = 0u;
                __syncthreads();
            }
            break;
#else
            if (histo) {
                // the last bucket up to which the children still fit a list: its upper edge is the new bound
                if (wave == 0) {
                    constexpr int PER = BN_BUCKETS / 64;
                    u32 mine = 0;
                    for (int i = 0; i < PER; i++) mine += hist[lane * PER + i];
                    u32 incl = mine;
                    for (int d = 1; d < 64; d <<= 1) {
                        const u32 t = (u32)__shfl_up((int)incl, d);
                        if (lane >= d) incl += t;
                    }
                    u32 run = incl - mine;
                    const u32 fillTo = (u32)(FCAP * KB_BNB_FILL / 16) < (u32)(KB_BNB_FILLK * k) ? (u32)(FCAP * KB_BNB_FILL / 16) : (u32)(KB_BNB_FILLK * k);
                    int fit = 0;  // buckets of this lane's share that still fit
                    for (int i = 0; i < PER; i++) {
                        run += hist[lane * PER + i];
                        if (run <= fillTo) fit = i + 1;
                    }
                    const u64 full = __ballot(fit == PER);
                    const int firstShort = full == ~0ull ? 64 : __ffsll((long long)~full) - 1;
                    const int nb = firstShort * PER + __shfl(fit, firstShort < 64 ? firstShort : 63);
                    if (lane == 0) ctl->limit = firstShort == 64 ? U : base + (double)nb * ((U - base) / (double)BN_BUCKETS);
                }
                __syncthreads();
                const double nu = ctl->limit;
                for (int i = tid; i < BN_BUCKETS; i += NT) hist[i] = 0u;
                if (tid == 0) lvlN[level + 1] = 0;
                __syncthreads();
                if (!(nu > base) || !(nu < U)) { stop = true; break; }  // (one bucket alone overfills the list)
                U = nu;
                scale = (double)BN_BUCKETS / U;
                Uprune = U * (1.0 + 1e-12);
                tight++;
                histo = false;
                continue;  // (this level again, under the new bound)
            }
            if (!last && lvlN[level + 1] > FCAP) {
                if (mode != 0 || base < 0.0 || tight >= 8) { stop = true; break; }
                histo = true;  // (mode 0 before the last level: the histogram is all zero)
                continue;
            }
            break;
#endif
            }
            if (stop) {
                if (tid == 0) ctl->abort = 1;
                break;
            }
            BEntry *T = A; A = B; B = T;
        }
        if (mode == 0) {
            mn = wave_min_f64(mn);
            if (lane == 0 && mn < INF) atomicMin(&ctl->minBits, (unsigned long long)__double_as_longlong(mn));
            if (cnt) atomicAdd(&ctl->count, cnt);
        }
        __syncthreads();
        return U;
    };

    // ---- the bound search: the smallest workable U with k assignments below it --------------------------------------------
    double U = gsum < INF ? gsum * KB_BNB_START : BN_GATE / 32.0;
    if (U < BN_GATE / 32.0) U = BN_GATE / 32.0;
    if (U > Umax) U = Umax;
    double Ulo = 0.0, Uhi = INF;  // a pass at Ulo counted < k; a pass at Uhi did not fit the lists
    int count = 0, cPrev = 0;
    double uPrev = 0.0;
    bool found = false;
    // (diagnostics, kbest_set_profile_buffer: [0] passes, [1] passes that did not fit, [3] candidates, [5] cycles)
    unsigned long long dPass = 0, dOver = 0;
    const unsigned long long dT0 = __builtin_readcyclecounter();
    unsigned long long dT1 = 0, dT2 = 0;
    for (int it = 0; it < 40; it++) {
        const double Uin = U;
        U = walk(Uin, 0, (KB_BNB_LOWER == 1 || (KB_BNB_LOWER == 2 && NT == 256)) ? Ulo : -1.0);  // (may come back lowered: the count is exact for what comes back)
        const bool over = ctl->abort != 0;
        const bool lowered = U < Uin;
        count = ctl->count;
        dPass++;
        dOver += (over || lowered) ? 1 : 0;
        __syncthreads();
        if (over) {
            Uhi = Uin;
            if (!(Uhi - Ulo > 1e-9 * Uhi)) break;
            U = 0.5 * (Ulo + Uhi);
            continue;
        }
        if (lowered) Uhi = Uin < Uhi ? Uin : Uhi;
        if (count >= k || (!lowered && U >= Umax)) { found = true; break; }
        if (lowered && !(U - Ulo > 1e-7 * U)) break;  // (the largest bound that fits the lists has fewer than k below it)
        // the number of assignments below a bound grows like a power of it: the exponent from the last two passes that counted
        // (4 without them), the next bound aimed at 1.5 k assignments, a factor between 1.05 and 2
        const double fmax = NT >= 1024 ? KB_BNB_FMAX_LONE : KB_BNB_FMAX;
        double f = fmax;
        if (count > 0) {
            double pw = 4.0;
            if (cPrev > 0 && count > cPrev && U > uPrev) {
                pw = log((double)count / (double)cPrev) / log(U / uPrev);
                pw = pw < 2.0 ? 2.0 : (pw > 12.0 ? 12.0 : pw);
            }
            f = exp(log(KB_BNB_AIM * (double)k / (double)count) / pw);
            f = f < 1.05 ? 1.05 : (f > fmax ? fmax : f);
            cPrev = count;
            uPrev = U;
        }
        Ulo = U;
        double nxt = U * f;
        if (Uhi < INF && nxt >= Uhi) nxt = 0.5 * (U + Uhi);
        if (nxt > Umax) nxt = Umax;
        if (!(nxt > U)) break;
        U = nxt;
    }
    if (!found) {  // (no bound with k assignments below it fits the lists -- masses of equal or nearly equal gains: the enumeration kernels)
        if (p.prof && tid == 0) {
            unsigned long long *d = p.prof + (long long)b * 16;
            d[0] = dPass; d[1] = dOver; d[6] = 1; d[7] = (unsigned long long)count;
            d[8] = (unsigned long long)__double_as_longlong(Ulo); d[9] = (unsigned long long)__double_as_longlong(Uhi);
        }
        if (tid == 0) p.nf[b] = -2;
        signal_done();
        return;
    }
    if (count == 0) {  // infeasible: kBest2D returns 0 (cpp:588-593)
        if (tid == 0) p.nf[b] = 0;
        signal_done();
        return;
    }
    dT1 = __builtin_readcyclecounter();
    const double best = __longlong_as_double((long long)ctl->minBits);  // gainBest[0] (CDelta = 0 on a conditioned matrix)
    const double cutG = best + p.cutoff;                                  // cpp:681
    // the first bucket at which the cumulated count reaches k: its upper edge bounds the k-th gain
    if (tid == 0) ctl->bStar = BN_BUCKETS - 1;
    __syncthreads();
    if (wave == 0 && count >= k) {
        constexpr int PER = BN_BUCKETS / 64;
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
                if ((int)run >= k) { ctl->bStar = lane * PER + i; break; }
            }
        }
    }
    __syncthreads();
    {
        const int bStar = ctl->bStar;
        double E = (bStar < BN_BUCKETS - 1) ? (double)(bStar + 1) * (U / (double)BN_BUCKETS) * (1.0 + 1e-12) : U;
        if (E > U) E = U;
        if (cutG < E) E = cutG;  // (what lies beyond the cutoff is never emitted: cpp:709-719)
        if (KB_BNB_KEEP && count <= CAP) {
            // The pass that found k assignments below its bound kept every one of them (count <= CAP: none was dropped): what the
            // collecting walk would bring -- the assignments with gain <= E -- is a subset of the list.  Compacted in place: the kept
            // entries beyond the first K positions move into the holes among the first K (the rank sort does not mind the order).
            int *movers = rankA, *holes = rankA + CAP / 2;  // (at most min(K, count - K) <= CAP / 2 of each)
            int *cnt3 = lvlN;                               // [0] kept, [1] movers, [2] holes (the level counts are dead)
            if (tid < 3) cnt3[tid] = 0;
            __syncthreads();
            int kept = 0;
            for (int e = tid; e < count; e += NT) kept += (candG[e] <= E) ? 1 : 0;
            if (kept) atomicAdd(&cnt3[0], kept);
            __syncthreads();
            const int K = cnt3[0];
            for (int e = tid; e < count; e += NT) {
                const bool keep = candG[e] <= E;
                if (keep && e >= K) movers[atomicAdd(&cnt3[1], 1)] = e;
                if (!keep && e < K) holes[atomicAdd(&cnt3[2], 1)] = e;
            }
            __syncthreads();
            for (int i = tid; i < cnt3[1]; i += NT) {
                const int from = movers[i], to = holes[i];
                candG[to] = candG[from]; candLo[to] = candLo[from]; candHi[to] = candHi[from];
            }
            if (tid == 0) { ctl->listN = K; ctl->abort = 0; }
            __syncthreads();
        } else {
            if (tid == 0) ctl->listN = 0;
            __syncthreads();
            walk(E, 1, -1.0, KB_BNB_KEEP ? M - 1 : 0);  // (KEEP builds: the last level again, over the frontier the counting pass left)
        }
    }
    dT2 = __builtin_readcyclecounter();
    const int n = ctl->listN;
    if (n > CAP || ctl->abort) {
        if (tid == 0) p.nf[b] = -2;
        signal_done();
        return;
    }
    // ---- rank sort by (gain, rows); several threads share an element's comparisons ---------------------------------------
    for (int i = tid; i < CAP; i += NT) rankA[i] = 0;
    __syncthreads();
    {
        const int per = (n > 0 && NT / n > 0) ? NT / n : 1;
        for (int e0 = 0; e0 < n; e0 += NT / per) {
            const int e = e0 + tid / per, part = tid % per;
            if (e < n && tid / per < NT / per) {
                const double g = candG[e];
                const u64 lo = candLo[e], hi = candHi[e];
                // gains below mine, and gains equal to mine (myself among them, once, in one part): the rows break exact ties only
                int rk = 0, eq = 0, j = part;
                for (; j + 7 * per < n; j += 8 * per) {  // (eight reads in flight)
                    double g2[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) g2[q] = candG[j + q * per];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        rk += g2[q] < g ? 1 : 0;
                        eq += g2[q] == g ? 1 : 0;
                    }
                }
                for (; j < n; j += per) {
                    const double g2 = candG[j];
                    rk += g2 < g ? 1 : 0;
                    eq += g2 == g ? 1 : 0;
                }
                if (eq > ((e % per) == part ? 1 : 0)) {  // (some OTHER candidate has my gain, bit for bit: rare)
                    for (j = part; j < n; j += per)
                        if (candG[j] == g) {
                            // the canonical order of exact ties (kbest_ties.h): rows of column 0 first -- byte 0 of the low word
                            const u64 a0 = __builtin_bswap64(candLo[j]), a1 = __builtin_bswap64(candHi[j]);
                            const u64 m0 = __builtin_bswap64(lo), m1 = __builtin_bswap64(hi);
                            if (a0 < m0 || (a0 == m0 && a1 < m1)) rk++;
                        }
                }
                if (rk) atomicAdd(&rankA[e], rk);
            }
        }
    }
    for (int e = tid; e < n; e += NT)  // cpp:709-719: solutions are counted while not beyond gainBest[0] + cutoff
        if (!(candG[e] > cutG)) atomicAdd(&ctl->nWithin, 1);
    __syncthreads();
    const unsigned long long dE1 = __builtin_readcyclecounter();
    int nf = ctl->nWithin;
    nf = nf < k ? nf : k;
    for (int e = tid; e < n; e += NT) {
        const int rk = rankA[e];
        if (rk < nf) {
            solG[rk] = candG[e];
            const u64 lo = candLo[e], hi = candHi[e];
            for (int c = 0; c < M; c++) {  // (every unassigned measurement's own row counts as "no landmark": row nLc, :634-637)
                const int r = row_of(lo, hi, c);
                rTab[rk * 16 + c] = (unsigned char)(r >= nLc ? nLc : r);
            }
        }
    }
    __syncthreads();
    // exact ties: every assignment up to the k-th gain's bucket is in the list, ordered canonically: the k kept are the
    // lexicographically first of their gain level; the flag says that there was a choice (as in kbest_tiny.hip)
    if (p.tieFlags && nf == k)
        for (int e = tid; e < n; e += NT)
            if (rankA[e] == k && candG[e] == solG[k - 1] && !(candG[e] > cutG)) p.tieFlags[b] = KBEST_TIE_BOUNDARY | KBEST_TIE_RESOLVED;
    const unsigned long long dE2 = __builtin_readcyclecounter();
    // ---- the weights (assignment.cpp:616-648), as in kbest_small.hip ------------------------------------------------------
    for (int s = tid; s < nf; s += NT) {
        const double g = solG[s];
        wts[s] = (p.gate && !(best + BN_GATE > g)) ? 0.0 : exp(best - g);  // :622-626 (0.0: skipped -- x + 0.0 is x, bit for bit, for the x >= +0.0 here)
    }
    __syncthreads();
    const unsigned long long dE3 = __builtin_readcyclecounter();
    // probs[col][row] = the weights of the solutions that give `row` to `col`, added in ascending order of the solutions, over the
    // total of all weights added the same way (:633-643).  A wave per column, a lane per row: the wave goes through the solutions
    // in order (weight and row: one address for all lanes), the lane whose row it is adds -- the additions of the reference's
    // loop and no others; every wave adds up the total as well.  (As many columns to a wave as have room in its 64 lanes.)
    const int nAcc = M * (nLc + 1);
    double *accS = reinterpret_cast<double *>(lastB);  // [M][nLc + 1] (the candidates are dead)
    const int nRowA = nLc + 1;
    const int cpw = 64 / nRowA;  // columns per wave (>= 1: nLc + 1 <= 64 - M + 1)
    const int myCl = lane / nRowA, myR = lane - myCl * nRowA;
    for (int c0w = wave * cpw; c0w < M; c0w += NWV * cpw) {
        const int myC = c0w + myCl;
        const bool mineOn = myCl < cpw && myC < M;
        double acc = 0.0, total2 = 0.0;
        const unsigned char *rp = rTab + (mineOn ? myC : 0);
        const int want = mineOn ? myR : -1;
        int s2 = 0;
        for (; s2 + 8 <= nf; s2 += 8) {
            double w[8];
            int r[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                w[q] = wts[s2 + q];
                r[q] = rp[(s2 + q) * 16];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double a2 = acc + w[q];
                total2 = total2 + w[q];
                acc = (r[q] == want) ? a2 : acc;
            }
        }
        for (; s2 < nf; s2++) {
            const double w = wts[s2];
            const double a2 = acc + w;
            total2 = total2 + w;
            acc = ((int)rp[s2 * 16] == want) ? a2 : acc;
        }
        if (mineOn) accS[myC * nRowA + myR] = acc;
        if (c0w == 0 && lane == 0) ctl->limit = total2;
    }
    __syncthreads();
    {
        const double norm = 1.0 / ctl->limit;  // :643
        for (int i = tid; i < nAcc; i += NT) {
            const int accC = i / (nLc + 1), accR = i - accC * (nLc + 1);
            // scatter back to the caller's landmark numbering (getAssignmentProbs, assignment.cpp:68-74)
            const int ro = (accR >= nLc) ? nLout : (int)rowIdx[accR];
            probOut[accC * (nLout + 1) + ro] = accS[i] * norm;
        }
    }
    if (tid == 0) p.nf[b] = nf;
    if (p.prof && tid == 0) {
        unsigned long long *d = p.prof + (long long)b * 16;
        d[0] = dPass; d[1] = dOver; d[3] = (unsigned long long)n;
        d[5] = __builtin_readcyclecounter() - dT0;
        d[6] = dE1 - dT2; d[7] = dE2 - dE1; d[8] = dE3 - dE2; d[9] = __builtin_readcyclecounter() - dE3; d[10] = dT0 - dStart; d[2] = dSA - dStart; d[4] = dSB - dSA; d[14] = dSC - dSB; d[15] = dSD - dSC; d[11] = dT1 - dT0; d[12] = dT2 - dT1; d[13] = __builtin_readcyclecounter() - dT2;
    }
    signal_done();
}

int bnb_lds_bytes(int k, int nThreads, int maxRow, int maxCol)
{
    (void)k;
    const int tileCol = maxCol < BNB_MAX_COL ? maxCol : BNB_MAX_COL;
    const int tileRow = (maxRow > 0 && maxRow < BNB_MAX_ROW) ? maxRow : BNB_MAX_ROW;
    int o = tileCol * (tileRow + 1) * 8 + BNB_MAX_COL * 8 + 80 + 8 + BNB_MAX_ROW * 2 + BN_BUCKETS * 4 + ((tileCol * tileRow + 7) & ~7) + BNB_MAX_COL +
            (BNB_MAX_COL + 1) * 4 + 12 + BNB_MAX_COL + BNB_MAX_COL * 40 + 2 * bnb_frontier_cap(nThreads) * 32;
    return (o + 31) & ~15;
}

template <int NT>
static hipError_t launch_bnb_nt(const SmallParams &p, int B, hipStream_t stream)
{
    const int lds = bnb_lds_bytes(p.kTab, NT, p.bnbRow, p.maxCol);
    static std::atomic<int> granted[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > granted[dev & 15].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_bnb_kernel<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        granted[dev & 15].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((kbest_bnb_kernel<NT>), dim3(B), dim3(NT), lds, stream, p);
    return hipGetLastError();
}

// many = the batch fills the chip: smaller workgroups
hipError_t launch_kbest_bnb(const SmallParams &p, int B, bool many, hipStream_t stream)
{
    return many ? launch_bnb_nt<256>(p, B, stream) : launch_bnb_nt<1024>(p, B, stream);
}

}  // namespace kb
