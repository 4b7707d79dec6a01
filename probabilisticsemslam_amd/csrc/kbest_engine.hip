// kbest_engine.hip -- MI355X (gfx950) k-best assignment engine: device kernels.
//
// One workgroup per cost matrix, persistent over all k sweeps of Murty's
// enumeration.  What the reference does with one heap of fully solved child
// hypotheses (shortestPathCPP.cpp:455-532, 571-644) is restructured for the
// GPU as
//
//   * cost tile in LDS (column stride D|1 so both "lane = row, fixed column"
//     and "lane = column, fixed row" walks are bank-conflict free), read from
//     HBM exactly once, shifted on the fly (makeCostMatrixSafe, cpp:534-569);
//   * one wavefront per child, lane = row: the reduced-cost scan of
//     shortestPathUpdateCPP (cpp:307-325) is one LDS read + three fp64 adds per
//     lane, the arg-min is a DPP min-reduction + ballot/ff1 (lowest row index
//     wins, as the ascending Row2Scan walk of the reference does), the
//     scanned / candidate / forbidden row sets are 64-bit scalar masks;
//   * first-step filter: the minimum first-step reduced cost of ALL children
//     of a node is computed in one vector pass (lane = child column); children
//     whose first Dijkstra step would already exceed the bound never get a wave;
//   * early termination: Dijkstra's running distance `delta` is a lower
//     bound of the child's gain (parent gain + delta), so a child is dropped
//     as soon as that bound exceeds the current k-th best candidate (with a
//     safety margin far above rounding error);
//   * bounded pool: only the (k - emitted) smallest candidates can ever be
//     output, so the queue is a sorted LDS array of at most k entries, merged
//     by rank each round;
//   * surviving children are finished like the reference finishes them (flip,
//     dual update, exact gain) and their whole state is kept in HBM; when the
//     state slots run out a child stays lazy -- (gain, parent, column) only --
//     and is re-solved from its parent's state if it is ever selected: the
//     same function of the same inputs, hence the same result;
//   * batched frontier: per round the first `spec` unsplit candidates of the
//     pool are split together (speculatively, except the minimum), which turns
//     199 sequential sweeps into a few dozen rounds of wave-parallel work.
//
// Results are identical to the reference for every hypothesis that is output:
// the same assignments in the same order, gains bit-identical (serial
// column-order sum, calcGain cpp:59-80), duals of emitted hypotheses
// bit-identical (updateDualAndAugment cpp:82-117 evaluated in the same order).
//
// fp64 add/sub/compare only -- no MFMA (nothing here is a contraction).
// Compiled WITHOUT fast-math: ((delta + C) - u) - v must not be reassociated.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "kbest_engine.h"
#include "kbest_wave.h"
#include "kbest_lap.h"

namespace kb {


// ------------------------------------------------------------------ the kernel
//
// Batched-frontier form of Murty's loop (kBest2D cpp:607-634 + split
// cpp:455-532).  Per round:
//   B  all children of all freshly solved nodes are solved for their gain by
//      all waves (dynamic work queue), with early termination; survivors are
//      appended to the fresh list;
//   C  the fresh list is merged by rank into the sorted pool, which is cut to
//      the (k - emitted) candidates that can still be output;
//   A  the first `spec` not-yet-solved candidates of the pool are re-solved IN
//      PARALLEL, one wave each, from their parents' saved states (exactly the
//      reference's shortestPathUpdateCPP on that (parent, column));
//   D  the pool head is emitted while it is solved: a solved head is certain
//      to be the next-best hypothesis once every hypothesis emitted before it
//      has been split, so a head solved only in this round (not split yet) is
//      emitted too but ends the run.
// With spec = 1 this is the reference's order of operations exactly (pop the
// minimum, split it, emit the new top).  With spec > 1 later candidates are
// split speculatively; the pool always holds a partition of the not yet
// emitted assignments, so the emitted sequence -- assignments in increasing
// gain -- is unchanged (tie-free inputs).
// ceil(65536 / d), d = 1..16: (x * RCP16[d]) >> 16 == x / d for 0 <= x < 4096
__constant__ const int RCP16[17] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096};

struct Ctrl {
    double cdelta;      // CDelta * numCol (cpp:583)
    double cutoffGain;  // workMem.cutoffGain (cpp:681/684)
    double cmax;        // largest finite shifted cost (scale of the safety margin)
    double gain0u;      // gainBest[0]
    int nq;        // end of the valid pool range
    int head;      // start of it (entries before head were emitted)
    int emitted;   // output slots filled so far
    int stop;      // 1: finished   2: internal error   3: infeasible root
    int pushed;
    int nsel;      // nodes to split in this round
    int nextItem;  // work queue of phase B
    int nextSid;   // next free state slot of the lazy region [0, lazyStates)
    int nextEager; // next free state slot of the eager region [lazyStates, statesPerProblem)
    int nFresh;    // surviving children appended this round
    int nSurv;     // children that passed the first-step filter this round (queued from the front)
    int nSurvBack; //   ... and those queued from the back
    short selIdx[16];          // pool index of each node selected in the last A phase (they are split in the next B)
    unsigned short selSid[16]; // and its state slot
    int partsDone[16];         // waves that have finished their part of a node's first-step filter (the last one compacts)
    double t0;                 // a-priori threshold on the k-th best gain (apriori_threshold), +inf when unknown
    int outDone;               // output slots whose row4col / col4row tables have been written
    int outTicket;             // work queue of the output writes of this round
    double tShared;            // split launches: the smallest threshold any share of this matrix has published (+inf: none)
    int relayCut;              // relay launches: this workgroup hands the matrix over once so many solutions are out (else INT_MAX)
    int relayRound;            // ... and the round the next piece starts with (part of the LDS image)
};
static_assert(sizeof(Ctrl) <= 240, "Ctrl must fit the LDS slot reserved by lds_layout");

// Optimistic bounds and re-split tickets.  The pool's threshold T (the gain of its last entry once it holds k - emitted
// candidates) is a VALID bound, but a loose one for most of the run: half of the children that complete under it never make
// the k best (tests/dev/proto_tickets.cpp: 4 096 x 32x32, k = 200: 498 completions per matrix for 199 that are needed).  A node is
// therefore split against an OPTIMISTIC bound Tg <= T -- a quantile of the pool's candidates -- and the children that die
// between the two are not lost: the smallest lower bound among them (first + last arc at the filter, the settled distance at
// which a search was given up) is kept as the node's TICKET.  Tickets are a small sorted list beside the pool: a candidate
// is only emitted while its gain is below the smallest ticket key, and a ticket that reaches the front of the union is
// selected like a candidate -- its node is split AGAIN against a higher bound, skipping the children that completed before
// (`done` mask, kept with the node's saved state).  Whatever Tg is, the enumeration stays exact: an optimistic guess costs
// a re-split, never a result.  Measured on the host model: completions 498 -> 292, Dijkstra steps -24 %, 3 tickets per
// matrix, rounds + 2 %.
struct Opt {
    u64 defKey[16];     // per node of this round: smallest lower bound among its deferred children (ordered key of the shifted gain; ~0: none)
    u64 done[16];       // per node: positions (columns in the enumeration's order) whose child has completed, now or in an earlier split
    double bAbs[16];    // per node: the (absolute, shifted-gain) optimistic bound it is split against, set when it is selected
    double gRoot;       // the optimum's shifted gain
    int nT;             // tickets in the list
    u32 selTicket;      // nodes 0 .. selTicket - 1 of this round are tickets (re-splits): the first selTicket entries of the list
    int anyFinite;      // some node of this round has a finite optimistic bound (the filter's last-arc pass is worth running)
    double TK[OPT_TICKETS];          // ticket keys, ascending
    unsigned short TS[OPT_TICKETS];  // state slot of each ticket's node
};
static_assert(sizeof(Opt) <= OPT_BYTES, "Opt must fit the LDS slot reserved by lds_layout");
constexpr int OPT_TSEL = 4;  // tickets that can be selected in one round (the first ones of the list)


// A-priori threshold on the k-th best gain, computed once per problem when the root's children are solved (round 1).
// Every child of the root differs from the optimum by ONE alternating path ("atom": extra cost d_i, rows moved m_i).
// Atoms that move disjoint rows touch disjoint columns, so applying several of them at once is again an assignment, it
// costs the sum, and different atom sets give different assignments (the symmetric difference with the optimum
// decomposes uniquely into its paths).  Singles, pairs, triples of the 16 cheapest and quadruples of the 8 cheapest
// atoms are a few thousand KNOWN assignments: the (k-1)-th smallest of their costs, plus the optimum, bounds the k-th
// best gain from above -- 1.7-3.5x the true gap on 64x64, k = 200 -- while the pool has no threshold at all yet
// (round 1 would otherwise complete every child of 8 hypotheses) and only a loose one for some rounds after.
// The (k-1)-th smallest is bracketed by bisection on the value: every thread keeps its share of the combinations in
// registers and counts those <= x; the upper end of the bracket is a valid bound whatever the precision.
// A separate function on purpose: its registers must not count against the round loop's.
constexpr int T0_SCRATCH = 160;  // u64 words of LDS scratch: sorted costs [64], masks [64], one counter per bisection step [32 x i32]
constexpr int T0_EXTRA = 160;    // atoms learnt in round 1 (HBM, behind the root's)
// words of the atoms area in HBM: [2c] cost bits, [2c+1] rows moved, of the root's child on column c (c < D); [2D] the
// root's gain; [2D+1] number of learnt atoms; then T0_EXTRA x (cost bits, rows moved)
__host__ __device__ inline int t0_area_bytes(int D) { return 8 * (2 * D + 2 + 2 * T0_EXTRA); }
// phase 0 (top of round 1): the atoms are the root's children.  phase 1 (top of round 2): the children completed in
// round 1 that differ from the optimum by ONE path too (they mostly are the second-best way through a region that one of
// the eight best atoms already covers -- the alternatives that the true k best are full of) join them; the 64 cheapest
// of all are used.  They are distinct assignments (pool entries of one enumeration), each a single connected change of
// the optimum, so the counting argument is the same: 1.2-1.8x the true gap instead of 1.7-3.5x.
// `scratchWords`: u64 words of LDS available (the fresh list's gains); candidates beyond it are dropped (fewer known
// assignments: a looser bound, still a bound).
template <int NW>
__device__ __attribute__((noinline)) void apriori_threshold(u64 *scratch, const u64 *atoms, double *t0Out, int D, int k,
                                                            int phase, int scratchWords)
{
    constexpr int NT = NW * 64, SLOTS = (4096 + NT - 1) / NT;  // NW >= 8: at most 8 grid slots per thread and kind
    const double INF = d_inf();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *sd = reinterpret_cast<double *>(scratch);  // atoms sorted by cost
    u64 *sm = scratch + 64;                            // their row masks
    int *cnt = reinterpret_cast<int *>(scratch + 128);
    const double gRoot = __longlong_as_double((long long)atoms[2 * D]);
    if (phase == 0) {
        if (wave == 0) {
            const double dl = lane < D ? __longlong_as_double((long long)atoms[2 * lane]) : INF;
            const u64 ml = lane < D ? atoms[2 * lane + 1] : 0ull;
            int rank = 0;
            for (int j = 0; j < 64; j++) {
                const double dj = readlane_f64(dl, j);
                rank += (dj < dl || (dj == dl && j < lane)) ? 1 : 0;
            }
            sd[rank] = dl;
            sm[rank] = ml;
        }
    } else {
        // candidates: every atom that can still matter (cost below the current threshold's gap), compacted into LDS
        double *cd = reinterpret_cast<double *>(scratch + T0_SCRATCH);
        const int candCap = (scratchWords - T0_SCRATCH) / 2;
        u64 *cm = scratch + T0_SCRATCH + candCap;
        int *ncand = reinterpret_cast<int *>(scratch + 144);
        const double capGap = *t0Out - gRoot;  // (+inf when there is no threshold yet)
        // (the counter is bumped by L2 atomics; a plain load that hit a line cached before them could only read a SMALLER
        //  count -- fewer atoms, a looser bound, still a bound; the atoms themselves are plain stores of this workgroup's
        //  waves, coherent through the CU's vector L1)
        int nExtra = (int)(unsigned)atoms[2 * D + 1];
        nExtra = nExtra > T0_EXTRA ? T0_EXTRA : nExtra;
        if (tid < 64) { sd[tid] = INF; sm[tid] = 0ull; }
        if (tid == 0) *ncand = 0;
        __syncthreads();
        for (int i = tid; i < D + nExtra; i += NT) {
            const int w = i < D ? 2 * i : 2 * D + 2 + 2 * (i - D);
            const double dl = __longlong_as_double((long long)atoms[w]);
            if (dl < INF && dl <= capGap) {
                const int pos = atomicAdd(ncand, 1);
                if (pos < candCap) { cd[pos] = dl; cm[pos] = atoms[w + 1]; }
            }
        }
        __syncthreads();
        int nc = *ncand;
        nc = nc > candCap ? candCap : nc;
        for (int i = tid; i < nc; i += NT) {  // the 64 cheapest, sorted (ties by position)
            const double dl = cd[i];
            int rank = 0;
            for (int j = 0; j < nc; j++) {
                const double dj = cd[j];
                rank += (dj < dl || (dj == dl && j < i)) ? 1 : 0;
            }
            if (rank < 64) { sd[rank] = dl; sm[rank] = cm[i]; }
        }
    }
    if (tid < 32) cnt[tid] = 0;
    __syncthreads();
    const int nA = __popcll(__ballot(sd[lane] < INF));
    if (nA < 2) return;  // (uniform)
    // Every thread keeps its share of the combinations in registers as floats ROUNDED UP and counts them against x rounded
    // DOWN: a combination is counted only if it really is <= x, so the count can only be too small and the x at which it
    // reaches k - 1 is still an upper bound of the (k-1)-th smallest (half the registers of fp64 values: the function fits the
    // caller-saved registers and needs no stack).
    float val[3 * SLOTS + 1];
    val[3 * SLOTS] = __double2float_ru(tid < 64 ? sd[tid] : INF);  // singles
#pragma unroll
    for (int e = 0; e < SLOTS; e++) {
        const int idx = tid + e * NT;
        {   // pairs of all atoms
            const int i = idx >> 6, j = idx & 63;
            double x = INF;
            if (idx < 4096 && i < j && j < nA && (sm[i] & sm[j]) == 0ull) x = sd[i] + sd[j];
            val[e] = __double2float_ru(x);
        }
        {   // triples of the 16 cheapest
            const int i = idx >> 8, j = (idx >> 4) & 15, l = idx & 15;
            double x = INF;
            if (idx < 4096 && i < j && j < l && l < nA) {
                const u64 mi = sm[i], mj = sm[j], ml = sm[l];
                if (((mi & mj) | (mi & ml) | (mj & ml)) == 0ull) x = (sd[i] + sd[j]) + sd[l];
            }
            val[SLOTS + e] = __double2float_ru(x);
        }
        {   // quadruples of the 8 cheapest
            const int i = idx >> 9, j = (idx >> 6) & 7, l = (idx >> 3) & 7, q = idx & 7;
            double x = INF;
            if (idx < 4096 && i < j && j < l && l < q && q < nA) {
                const u64 mi = sm[i], mj = sm[j], ml = sm[l], mq = sm[q];
                if (((mi & mj) | (mi & ml) | (mi & mq) | (mj & ml) | (mj & mq) | (ml & mq)) == 0ull)
                    x = ((sd[i] + sd[j]) + sd[l]) + sd[q];
            }
            val[2 * SLOTS + e] = __double2float_ru(x);
        }
    }
    auto total_le = [&](double x, int step) -> int {  // block-wide number of combinations <= x (one barrier); never too large
        const float xf = __double2float_rd(x);
        int n = 0;
#pragma unroll
        for (int e = 0; e <= 3 * SLOTS; e++) n += (val[e] <= xf) ? 1 : 0;
        int w = 0;
#pragma unroll
        for (int bit = 0; bit < 5; bit++) w += __popcll(__ballot((n >> bit) & 1)) << bit;  // n <= 25
        if (lane == 0 && w) atomicAdd(&cnt[step], w);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(cnt[step]);
    };
    // bracket: the answer is of the order of the cheapest atoms' sums -- start at twice the 16th cheapest atom and double
    // until k - 1 combinations are below (every single and every pair is <= twice the dearest atom), then 8 bisections
    const double top = 2.0 * sd[nA - 1];
    double hi = 2.0 * sd[nA > 16 ? 15 : nA - 1], lo = 0.0;
    int step = 0;
    for (;;) {
        if (hi > top) hi = top;
        if (total_le(hi, step++) >= k - 1) break;
        if (hi >= top || step >= 12) return;  // fewer than k - 1 known assignments: no threshold (uniform)
        lo = hi;
        hi = 2.0 * hi;
    }
    for (int it = 0; it < 8; it++) {
        const double mid = 0.5 * (lo + hi);
        if (total_le(mid, step++) >= k - 1) hi = mid; else lo = mid;
    }
    if (tid == 0 && gRoot + hi < *t0Out) *t0Out = gRoot + hi;
}

// pool entry: gain (fp64), meta (u32: column | parent state << 8 | flags), own state slot (u16)
constexpr unsigned short SID_NONE = 0xFFFFu;  // no saved state: re-solve from the parent when selected
constexpr u32 META_SPLIT = 0x80000000u;        // children already generated and merged
constexpr u32 META_MASK = 0x00FFFFFFu;

// a solved hypothesis waiting to be split, in LDS
struct NodeRef {
    double *u, *v;
    double *gain;  // [0] shifted gain
    u64 *forb;
    int *info;     // [0] activeCol, [1] sid
    unsigned char *r4c, *c4r;
};

__device__ __forceinline__ NodeRef node_ref(unsigned char *base, int maxRow)
{
    NodeRef n;
    n.u = reinterpret_cast<double *>(base);
    n.v = n.u + maxRow;
    n.gain = n.v + maxRow;
    n.forb = reinterpret_cast<u64 *>(n.gain + 1);
    n.info = reinterpret_cast<int *>(n.forb + 1);
    n.r4c = reinterpret_cast<unsigned char *>(n.info + 2);
    n.c4r = n.r4c + maxRow;
    return n;
}

// Register budget: 6 waves per SIMD (80 VGPRs) lets three 8-wave (or six 4-wave) workgroups share a CU; without the
// bound the compiler settles at ~90-100 VGPRs and residency silently drops to two matrices per CU.
// (seven 4-wave workgroups per CU at 72 VGPRs -- the LDS would hold them -- were measured in round 4: 4 096 x 32x32 3.65 -> 3.72 ms,
//  six spills and more waves on an issue-bound kernel)
constexpr int min_waves_per_simd(int nw) { return nw <= 12 ? 6 : 4; }

// EPT: pool entries per thread held in registers across the in-place merge (k <= EPT * NW * 64)
// RELAY: the instantiation of relay launches (below).  Plain launches run an instantiation without a line of it: these kernels sit at
// the 80-register cliff, where every added line moves spills around -- and where the compiler has produced wrong code more than once
// (an 8-wave build WITH the resume block faulted on plain launches that never execute it; NOTES 10.6).
template <int NW, int EPT, bool RELAY>
__global__ void __launch_bounds__(NW * 64, min_waves_per_simd(NW)) kbest_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NW * 64;
    const double INF = d_inf();
    int tid = threadIdx.x;   // (not const: see the top of the round loop)
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // split > 1: ONE matrix is enumerated by `split` workgroups, each taking the root's children on columns c % split == share
    // (Murty's partition of the root is disjoint, cpp:455-532) into its own tables [share][matrix]; a k-way merge follows
    // (kbest_merge.hip).  They share an upper bound of the k-th best gain through sharedT (see the top of the round loop).
    // Relay (round 5): a matrix is enumerated by gridDim.y workgroups ONE AFTER THE OTHER -- a workgroup takes the matrix up to a
    // share of its k solutions, leaves its whole LDS in HBM and ends; the workgroup of the next piece (dispatched later, to
    // whatever slot is free then) picks it up.  A launch of a few generations ends with the slot whose matrices add up to the
    // most (25 % of a C4 launch's slot-time is idle, NOTES 10.3): pieces a fraction of a lifetime long let the slots even out.
    // Same rounds, same arithmetic, same results.
    // WHICH piece a workgroup enumerates is not its block index but the order in which the gridDim.y workgroups of its matrix
    // arrive: each CLAIMS the next piece of the matrix (one atomic on a word per matrix).  A piece therefore only ever waits for
    // pieces that have been claimed, i.e. whose workgroups are running: no deadlock whatever order the hardware dispatches
    // workgroups in (it dispatches them in ascending order, x fastest -- then block (x, j) claims piece j --, but HIP promises
    // nothing of the kind).
    const int blk = blockIdx.x;  // index of this matrix' work space and output tables
    int piece = 0;
    if constexpr (RELAY) {
        unsigned *word = reinterpret_cast<unsigned *>(smem);
        if (threadIdx.x == 0) {
            *word = atomicAdd(p.relayClaim + blk, 1u);
        }
        __syncthreads();
        piece = __builtin_amdgcn_readfirstlane((int)*word);
        __syncthreads();
    }
    const bool fresh = RELAY ? piece == 0 : true;
    // The words of a matrix -- pieces claimed, progress, workgroups gone -- are zero between launches: the LAST of the matrix'
    // gridDim.y workgroups to leave (whatever way it leaves) puts them back.  Nothing is cleared from outside and nothing depends on
    // a launch number: a captured launch can be replayed as it is (a memset node in front of the kernel did NOT work: the claim
    // words, updated by atomics, stay in L2, and the graph's memset went to memory past them).
    auto relay_depart = [&]() {
        if constexpr (RELAY) {
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this lane's store to the progress word, if any, has arrived)
                if (atomicAdd(p.relayGone + blk, 1u) == gridDim.y - 1u) {
                    __hip_atomic_store(p.relayFlag + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.relayClaim + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.relayGone + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    const int S = p.split > 1 ? p.split : 1;
    const int share = S > 1 ? blk / p.splitB : 0;
    const int b = S > 1 ? blk - share * p.splitB : blk;  // the matrix (inputs)
    const int rcStride = S > 1 ? S : p.rootColStride, rcOffset = S > 1 ? share : p.rootColOffset;
    const int N = p.nRow ? p.nRow[b] : p.maxRow;
    const int M = p.nCol ? p.nCol[b] : p.maxCol;
    const int k = p.k;
    if (fresh && p.tieGain && tid == 0) p.tieGain[blk] = __longlong_as_double(0x7ff8000000000000LL);  // no solution behind the tables (yet)
    if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol) {  // undefined in the reference
        if (fresh && tid == 0) p.nf[blk] = (M == 0 || N == 0) ? 0 : -1;  // (an empty frame: nothing to assign, nothing found)
        relay_depart();
        return;  // (every piece's workgroup sees the same shape: none of them waits)
    }
    // odd column stride of the LDS cost tile: row-wise (lane = row) and column-wise (lane = column) walks are
    // both bank-conflict free
    const int D = N, LDC = D | 1;
    const int spec = p.spec < NW ? p.spec : NW;  // nodes solved / split per round
    const Lds L = lds_layout(p.maxRow, k, p.spec, NW);
    double *Cs = reinterpret_cast<double *>(smem + L.offC);
    double *freshG = reinterpret_cast<double *>(smem + L.offFreshG);
    u32 *freshM = reinterpret_cast<u32 *>(smem + L.offFreshM);
    double *PG = reinterpret_cast<double *>(smem + L.offPoolG);
    u32 *PM = reinterpret_cast<u32 *>(smem + L.offPoolM);
    unsigned short *PS = reinterpret_cast<unsigned short *>(smem + L.offPoolS);
    // global: state slot of each output slot.  One whole number of 128-byte lines per matrix: the table is written
    // and later re-read by this workgroup through its CU's L1, and a line shared with a neighbouring matrix could
    // have been pulled into that L1 earlier by another workgroup of the same CU (stale bytes for our half).
    unsigned short *slotSid = p.slotSid + (long long)blk * slot_table_stride(k);
    double *red = freshG;  // cross-wave reduction scratch of phase 0
    unsigned short *surv = reinterpret_cast<unsigned short *>(smem + L.offSurv);
    // last-arc minima of the current nodes' children (high words).  Shares its LDS with the fresh list's meta words,
    // which are only written in B2 and consumed in the first half of the merge.
    u32 *lbIn = reinterpret_cast<u32 *>(smem + L.offFreshM);
    // first-step minima of the current nodes' children.  Shares its LDS with the fresh-gain list: the minima live
    // from the filter to the survivor compaction (B1), the fresh gains from B2 to the first half of the merge.
    u64 *lbKey = reinterpret_cast<u64 *>(smem + L.offFreshG);
    unsigned short *freshS = reinterpret_cast<unsigned short *>(smem + L.offFreshS);
    Ctrl *ctrl = reinterpret_cast<Ctrl *>(smem + L.offCtrl);
    Opt *opt = reinterpret_cast<Opt *>(smem + L.offOpt);
    // column order of the enumeration (phase 1b): colOf[position] = the reference's column, posOf = its inverse
    unsigned char *colOf = smem + L.offPerm, *posOf = colOf + 64;
    double *gainW = reinterpret_cast<double *>(smem + L.offGainW) + wave * 64;  // this wave's line of gain terms

    const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.ldRow * p.ldCol);
    const bool maximize = p.maximize != 0, useCut = p.useCutoff != 0;
    const bool prune = (p.flags & KBEST_FLAG_NO_PRUNE) == 0;
    // assign2D / shortestPathCPP semantics (cpp:119-238, 735-762): numCol augmentations on the rectangular problem,
    // unassigned rows stay -1, no zero-padded columns; NO_SHIFT: the matrix is already "safe" (shortestPathCPP is
    // called on workMem.C as it is)
    const bool rect = (p.flags & KBEST_FLAG_RECT_ROOT) != 0;
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const bool noShift = (p.flags & KBEST_FLAG_NO_SHIFT) != 0;
    const int gainCols = (p.gainCols > 0 && p.gainCols < M) ? p.gainCols : M;  // numCol4Gain (cpp:232)
    // rows on the zero-padded columns M .. D-1 are settled together in a child's search (dijkstra<>); 64 = never
    const int parkFrom = (!rect && !(p.flags & KBEST_FLAG_EXACT_ROOT) && M < N) ? M : 64;
    const int rl = lane < D ? lane : D - 1;
    const u64 allRows = (D >= 64) ? ~0ull : ((1ull << D) - 1ull);
    // the last state slot of the problem is not a hypothesis: it holds the root's gain and, per child of the root, its
    // distance from the optimum and the rows it moves (the a-priori threshold of round 1, below)
    const int nSlots = p.statesPerProblem;
    const int atomSlots = (t0_area_bytes(D) + (int)p.stateStride - 1) / (int)p.stateStride;
    const int maxSid = nSlots - atomSlots > p.lazyStates ? nSlots - atomSlots : nSlots;
#ifdef KB_PROFILE
    unsigned long long profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long profT0 = __builtin_readcyclecounter();
    // (behind the [B][16] stamps: [B][3] -- start / end of the workgroup on the 100 MHz wall clock, and the CU it ran on: the
    //  launch's timeline, tools/tail_profile.py)
    if (p.prof && threadIdx.x == 0) {
        const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID: SE, CU, SIMD, wave slot
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        // (relay launches: one record per workgroup = (piece, matrix), piece-major)
        const long long wg = (long long)piece * gridDim.x + blockIdx.x, nwg = (long long)gridDim.x * gridDim.y;
        p.prof[nwg * 16 + wg * 5] = wall_clock64();
        p.prof[nwg * 16 + wg * 5 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
#endif

    // ---- phase 0: makeCostMatrixSafe + zero padding (cpp:534-569, 582-585) --
    double cdelTile = 0.0;  // the shift of the tile (kept for phase 1b, which loads the columns again in another order)
    if (RELAY && !fresh) {
        // a later piece: wait for the workgroup before it (it claimed its piece earlier: it runs or has run), take over its LDS.
        // One lane polls with RELAXED loads (an acquire per poll would invalidate this CU's L1 under its other workgroups
        // every time), then ONE agent-scope acquire, waited for, in front of the barrier behind which everybody loads.
        if (tid == 0) {
            unsigned f;
            // (bounded: ~2^21 polls of ~1 us.  A piece only waits for workgroups that are RUNNING, so the bound is never met in a healthy
            //  launch; it is met when the words were left non-zero by a launch that faulted or was aborted -- then this matrix comes back
            //  with nf = -3, an engine error, instead of hanging the GPU; the host re-zeroes the words before the next relay launch.)
            unsigned polls = 0;
            for (;;) {
                f = __hip_atomic_load(p.relayFlag + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f >= (unsigned)piece) break;  // (pieces done: piece j hands over with j + 1; 15: the matrix is finished)
                if (++polls > (1u << 21) || piece >= (int)gridDim.y) { p.nf[blk] = -3; f = 15u; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            red[0] = __longlong_as_double((long long)f);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const unsigned f = (unsigned)uni32((int)(unsigned)__double_as_longlong(red[0]));  // (uniform: the return below is a scalar branch)
        if (f == 15u) { relay_depart(); return; }  // the matrix was finished by an earlier piece
        const uint4 *src = reinterpret_cast<const uint4 *>(p.relayBuf + (long long)blk * p.relayStride);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        __syncthreads();
        for (int i = tid; i < L.total / 16; i += NT) dst[i] = src[i];
#ifdef KB_PROFILE
        if (p.prof && threadIdx.x == 0) p.prof[(long long)gridDim.x * gridDim.y * 16 + ((long long)piece * gridDim.x + blockIdx.x) * 5 + 3] = wall_clock64();  // the image is in
#endif
    }
    if (fresh) {
        if (tid < 64) { colOf[tid] = (unsigned char)tid; posOf[tid] = (unsigned char)tid; }
        double mn = INF;  // min of C, or min of -C when maximising (max C = -min(-C), exact)
        for (int c = wave; c < M; c += NW)
            for (int r = lane; r < N; r += 64) {
                double x = Cg[r + (long long)c * N];
                x = maximize ? -x : x;
                mn = min_keep(mn, x);
            }
        mn = wave_min_f64(mn);
        if (lane == 0) red[wave] = mn;
        __syncthreads();
        mn = red[0];
        for (int w = 1; w < NW; w++) mn = min_keep(mn, red[w]);
        const double cdel = noShift ? 0.0 : (maximize ? -mn : mn);
        cdelTile = cdel;
        __syncthreads();
        double cm = 0.0;
        for (int c = wave; c < D; c += NW)
            for (int r = lane; r < N; r += 64) {
                double val = 0.0;
                if (c < M) {
                    const double x = Cg[r + (long long)c * N];
                    val = noShift ? x : (maximize ? (-x + cdel) : (x - cdel));  // cpp:558 / cpp:564
                    // inf - inf (e.g. an all-inf matrix) gives NaN; every comparison the reference makes with
                    // a NaN reduced cost is false (cpp:185, 314), i.e. the arc behaves exactly like +inf.  The
                    // integer-key compare of the Dijkstra step needs that made explicit.
                    if (val != val) val = INF;
                    if (val < INF && val > cm) cm = val;
                }
                Cs[r + c * LDC] = val;
            }
        cm = -wave_min_f64(-cm);
        if (lane == 0) red[wave] = cm;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; w++) cm = red[w] > cm ? red[w] : cm;
            ctrl->cmax = cm;
            ctrl->cdelta = cdel * (double)M;  // cpp:583
            ctrl->stop = 0;
            ctrl->pushed = 0;
            ctrl->nq = 0;
            ctrl->head = 0;
            ctrl->emitted = 0;
            ctrl->nsel = 0;
            ctrl->nextItem = 0;
            ctrl->nextSid = 1;
            ctrl->nextEager = p.lazyStates;
            ctrl->nFresh = 0;
            ctrl->nSurv = 0;
            ctrl->nSurvBack = 0;
            ctrl->selIdx[0] = -1;
            ctrl->selSid[0] = 0;
            ctrl->t0 = INF;
            ctrl->outDone = 0;
            ctrl->outTicket = 0;
            ctrl->tShared = INF;
            for (int i = 0; i < 16; i++) { ctrl->partsDone[i] = 0; opt->defKey[i] = ~0ull; opt->done[i] = 0ull; opt->bAbs[i] = INF; }
            opt->nT = 0;
            opt->selTicket = 0u;
            opt->anyFinite = 0;
            opt->gRoot = 0.0;
        }
        __syncthreads();
        for (int i = tid; i < spec * 64; i += NT) { lbKey[i] = ~0ull; lbIn[i] = ~0u; }  // (`red` is dead now)
    }

    unsigned char *stBase = p.states + (long long)blk * nSlots * p.stateStride;
    u64 *atoms = reinterpret_cast<u64 *>(stBase + (long long)maxSid * p.stateStride);  // [2c] delta bits, [2c+1] row mask; [2D] root gain
    // A-priori threshold (used from round 1 on): k - 1 known assignments besides the optimum bound the k-th best gain
    // from above.  Off where the enumeration is not the whole problem's (root-subtree sharding), where pushes are
    // counted or pruning is disabled, and where its scratch (the fresh list's LDS) would not fit.
    // (a caller's own root sharding wants the shard's OWN k best: no global bound there; the internal split merges, so a bound on
    //  the global k-th best from this share's atoms is exactly what it may prune with)
    const bool t0On = prune && k >= 3 && (p.rootColStride <= 1 || S > 1) && !(p.flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_T0)) &&
                      !rect && NW >= 8 && spec >= 3 && maxSid < nSlots;
    if (fresh && t0On && tid < D) atoms[2 * tid] = 0x7ff0000000000000ull;  // +inf: no such child (yet)
    if (fresh && t0On && tid == 0) atoms[2 * D + 1] = 0ull;                 // atoms learnt in round 1
    // the second, sharper threshold (atoms learnt in round 1) needs the "one connected change" test: permutation cycles,
    // i.e. square problems (on rectangular ones a change can be an open path through the unassigned rows)
    const bool t1On = t0On && N == M && spec * 64 >= T0_SCRATCH + 64;
    // optimistic bounds with re-split tickets (struct Opt): not where the reference's own order of operations is the point
    // (push counts, the unpruned mode, the assign2D entries) and not in split launches (their shares trade thresholds)
    // (compiled in only for the shapes of fewer than 8 waves -- the ones without a-priori thresholds, where it pays: kbest_capi.cpp;
    //  the 8 / 12 / 16-wave kernels carry none of it)
    constexpr bool OPT_SHAPE = NW < 8;
    // (and not with a cutoff: kBest2DCutoff's gate is a tight valid bound from the first round on -- on the KITTI-like frames the
    //  guesses bought 12 % fewer completions for 13 % more rounds, 0.97 -> 1.15 ms per 1 000 frames on this kernel)
    const bool optOn = OPT_SHAPE && prune && S == 1 && !rect && !useCut && k >= 3 && p.optRho0 < 1.0f &&
                       !(p.flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_OPT | KBEST_FLAG_EXACT_ROOT));
    const int offDone = ((18 * p.maxRow + 7) & ~7) + 24;  // saved state: columns whose child has been completed
    unsigned char *rootMap = smem + L.offRootMap;  // the optimum's col4row (lane = row)
    // saved hypothesis (HBM): u[D'] v[D'] (fp64) | row4col[D'] col4row[D'] (u8) | forb, gain, activeCol
    // exact ties (kbest_ties.h): the tables hold kTab = k - 1 slots, the k-th solution is enumerated for its gain only
    // (p.kTab is read where it is needed: a kernel argument is re-loaded from the argument segment, a local would be one more value
    //  live across the round loop)
#define kTab (p.kTab)
    const long long outBase = (long long)blk * kTab;
    const int DS = p.maxRow;
    const int offR4C = 16 * DS, offC4R = 17 * DS, offTail = (18 * DS + 7) & ~7;

    // Write a full hypothesis to state slot `sid`.
    auto store_state = [&](int sid, double u, double v, int r4c, int c4r, u64 forb, double gain, int activeCol) {
        unsigned char *st = stBase + (long long)sid * p.stateStride;
        double *sd = reinterpret_cast<double *>(st);
        if (lane < D) {
            sd[lane] = u;
            sd[DS + lane] = v;
            st[offR4C + lane] = (unsigned char)r4c;
            st[offC4R + lane] = (unsigned char)c4r;
        }
        if (lane == 0) {
            *reinterpret_cast<u64 *>(st + offTail) = forb;
            *reinterpret_cast<double *>(st + offTail + 8) = gain;
            *reinterpret_cast<int *>(st + offTail + 16) = activeCol;
        }
    };
    // Publish the hypothesis this wave holds (u already in nd.u) as node nd, and store it as state `sid`.
    auto save_node = [&](const NodeRef &nd, int sid, double v, int r4c, int c4r, u64 forb, double gain,
                         int activeCol) {
        store_state(sid, (lane < D) ? nd.u[lane] : 0.0, v, r4c, c4r, forb, gain, activeCol);
        if (lane < D) {
            nd.v[lane] = v;
            nd.r4c[lane] = (unsigned char)r4c;
            nd.c4r[lane] = (unsigned char)c4r;
        }
        if (lane == 0) {
            nd.gain[0] = gain;
            nd.forb[0] = forb;
            nd.info[0] = activeCol;
            nd.info[1] = sid;
        }
    };

    // ---- phase 1: root LAP (shortestPathCPP, cpp:119-238) on wave 0 -> node 0, state 0, slot 0 ----
    if (fresh && wave == 0) {
        const NodeRef nd = node_ref(smem + L.offNodes, p.maxRow);
        if (lane < D) nd.u[lane] = 0.0;
        double v = 0.0, spc, delta;
        int c4r = -1, r4c = -1, pred, sink = 0;
        u64 scanned;
        bool bad = false;
        const int nAug = rect ? M : D;
        u64 todo = (nAug >= 64) ? ~0ull : ((1ull << nAug) - 1ull);  // columns still to be augmented from
        if (!rect && !(p.flags & KBEST_FLAG_EXACT_ROOT)) {
            // Column reduction first (the initialisation of Jonker-Volgenant): u[c] = min of column c, and a row that is
            // the arg-min of exactly one column -- or of several: the lowest column wins -- is assigned to it.  All
            // reduced costs stay >= 0 and the assigned arcs are tight, so this is a valid starting point for the
            // shortest-augmenting-path steps below, which then run only from the columns left over (about a third of them
            // on dense problems instead of all: the 64 sequential augmentations of the reference's root, cpp:139-230,
            // cost 9 % of a 64x64 matrix' lifetime on one wave while eleven wait).  The optimal assignment the root ends
            // with is the same (it is unique for tie-free costs); its dual variables are another optimal pair than the
            // reference's, which no output depends on.  assign2D / shortestPathCPP (rect) keep the reference's own order:
            // their duals ARE an output.
            int *owner = reinterpret_cast<int *>(gainW);
            owner[lane] = 64;
            wave_fence();
            const int cc = lane < D ? lane : D - 1;
            const double *Ccol = Cs + cc * LDC;
            double m = INF;
            int am = 0;
            for (int r0 = 0; r0 < D; r0 += 4) {
                double x[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = Ccol[(r0 + i < D) ? r0 + i : D - 1];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool better = (r0 + i < D) & (x[i] < m);  // strict '<': the lowest row among equal minima
                    m = better ? x[i] : m;
                    am = better ? r0 + i : am;
                }
            }
            // (only the REAL columns take part: the zero-padded ones are dealt with after the loop below)
            const bool can = lane < M && m < INF;
            if (can) atomicMin(&owner[am], lane);
            wave_fence();
            if (lane < D) nd.u[lane] = can ? m : 0.0;
            r4c = (can && owner[am] == lane) ? am : -1;
            const int ow = owner[lane];
            c4r = (lane < D && ow < 64) ? ow : -1;
            wave_fence();
            if (M == D) {
                // Row reduction on top (square problems): a row that got no column takes v[r] = min over c of (C[r,c] - u[c])
                // -- its reduced costs stay >= 0, exactly: they are the same differences, and one of them becomes 0 -- and if
                // the column of that minimum is still free, the row is assigned to it (several rows on one column: the lowest
                // wins).  A third of the columns left over by the column reduction get a row this way and, more to the point,
                // the duals are closer to optimal: the augmentations below take 165 Dijkstra steps instead of 307 on a 64x64
                // matrix (59 / 107 at 32x32).  Not with zero-padded columns: their reduced cost 0 - 0 - v[r] must not go negative.
                double m2 = INF;
                int a2 = 0;
                const int rr = lane < D ? lane : D - 1;
                for (int c0 = 0; c0 < D; c0 += 4) {
                    double x[4], uu[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int cj = (c0 + i < D) ? c0 + i : D - 1;
                        x[i] = Cs[rr + cj * LDC];
                        uu[i] = nd.u[cj];
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double d = x[i] - uu[i];
                        const bool better = (c0 + i < D) & (d < m2);  // strict '<': the lowest column among equal minima
                        m2 = better ? d : m2;
                        a2 = better ? c0 + i : a2;
                    }
                }
                const bool want = lane < D && c4r < 0 && m2 < INF;
                if (want) v = m2;
                owner[lane] = 64;
                wave_fence();
                const bool colFree = __shfl(r4c, a2) < 0;  // (the column's own lane holds its row)
                if (want && colFree) atomicMin(&owner[a2], lane);
                wave_fence();
                if (want && colFree && owner[a2] == lane) c4r = a2;
                const int ow2 = owner[lane];
                if (lane < D && r4c < 0 && ow2 < 64) r4c = ow2;
                wave_fence();
            }
            todo &= __ballot(lane < M && r4c < 0);
        }
        while (todo) {
            const int c = __builtin_ctzll(todo);
            todo &= todo - 1;
            if (dijkstra<false>(Cs, LDC, nd.u, rl, lane, v, c4r, allRows, 0ull, c, INF, spc, pred, scanned, delta,
                                sink)) { bad = true; break; }
            dual_update_flip(nd.u, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, c);
        }
        if (!bad && !rect && !(p.flags & KBEST_FLAG_EXACT_ROOT) && M < D) {
            // The zero-padded columns (cpp:582-585).  With the real columns assigned, every row that is still free has
            // v = 0 (only scanned rows ever change v, and a scanned free row is the sink: unchanged) and every other row
            // v <= 0, so "padded column M + j <- the j-th free row, u = 0" is tight on the assigned arcs and leaves every
            // reduced cost 0 - 0 - v[r] >= 0: an optimal solution of the padded square problem without the N - M
            // augmentations over tied zero columns that the reference's root spends most of its steps on (334 of them on
            // a 28 x 10 frame).  Which padded column a free row sits on is immaterial.
            const u64 freeRows = __ballot(lane < D && c4r < 0);
            if (lane < D && c4r < 0) c4r = M + __popcll(freeRows & ((1ull << lane) - 1ull));
            int *slot = reinterpret_cast<int *>(gainW);  // row of each padded column, through LDS
            wave_fence();
            if (lane < D && c4r >= M) slot[c4r - M] = lane;
            wave_fence();
            if (lane >= M && lane < D) { r4c = slot[lane - M]; nd.u[lane] = 0.0; }
            wave_fence();
        }
        if (bad) {
            if (lane == 0) ctrl->stop = 3;
        } else {
            const double g = serial_gain(Cs, LDC, lane, r4c, gainCols, gainW);
            const u64 forb = bit64(__builtin_amdgcn_readlane(r4c, 0));  // cpp:235
            save_node(nd, 0, v, r4c, c4r, forb, g, 0);
            if (p.dualU && lane < M) p.dualU[(long long)b * p.ldCol + lane] = nd.u[lane];  // MurtyHyp::u, per column (hpp:53)
            if (p.dualV && lane < N) p.dualV[(long long)b * p.ldRow + lane] = v;           // MurtyHyp::v, per row (hpp:55)
            if (t0On && lane == 0) atoms[2 * D] = (u64)__double_as_longlong(g);
            if (lane == 0) opt->gRoot = g;
            if (t0On && lane < D) rootMap[lane] = (unsigned char)c4r;
            if (lane == 0) {
                ctrl->cutoffGain = maximize ? (g - p.cutoff) : (g + p.cutoff);          // cpp:681/684
                const double gu = maximize ? (-g + ctrl->cdelta) : (g + ctrl->cdelta);  // cpp:599-603
                ctrl->gain0u = gu;
                p.gain[outBase] = gu;
                slotSid[0] = 0;
                ctrl->emitted = 1;
                ctrl->nsel = 1;
                if (k == 1) ctrl->stop = 1;
            }
        }
    }
    __syncthreads();
    if (uni32(ctrl->stop) == 3) {  // infeasible: kBest2D returns 0 (cpp:588-593)
        if (tid == 0) { p.nf[blk] = 0; if (p.pushed) p.pushed[blk] = 0; }
        if (RELAY && tid == 0) __hip_atomic_store(p.relayFlag + blk, 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        relay_depart();
        return;
    }

    // ---- phase 1b: the column order of the enumeration -------------------------------------------------------------
    // Murty's partition (split, cpp:455-532) lets the child on column c keep the parent's rows on the columns BEFORE c.  The
    // k best assignments are the optimum with a few cheap changes; where those columns stand in the order decides how much
    // of a child is already fixed when it is searched.  With the columns that are DEAR to change first and the cheap ones
    // last, the children that carry the k best have nearly everything fixed (short searches, few candidate rows), and the
    // children on the dear columns die at the bound.  The order changes the tree, not the set of the k best nor their gains
    // -- those are sums over the reference's column order (serial_gain's `orig`), and the tables go out in that order too.
    // The key of column c: the exact cost of taking its row away from it with nothing else fixed (one search from the
    // root's duals, the waves share the columns).  Measured on the host first (tests/dev/colorder_probe.py, columns permuted
    // before the call): 1 024 x 64x64, k = 200 2.50 ms as given, 2.09 by a two-arc lower bound, 1.70 by this key; 4 096 x
    // 32x32 4.87 / 4.29 / 3.63.  Not where the reference's own order is part of the answer (push counts, unpruned mode, the
    // exact-root mode, the assign2D entries) and not in split launches.
    // (and not under root-subtree sharding: WHICH assignments the child on a column holds depends on the order of the columns --
    //  the columns before it are fixed --, so shards enumerated in different orders would not partition the problem; the shards
    //  of kbest_c.h are those of the reference's order, whatever kernel, launch shape or key arithmetic a rank uses)
    const bool reorder = !rect && prune && S == 1 && p.rootColStride <= 1 && M >= 3 && k >= 3 &&
                         !(p.flags & (KBEST_FLAG_EXACT_ROOT | KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_REORDER));
    if (fresh && reorder) {
        double *key = reinterpret_cast<double *>(smem + L.offGainW);  // (wave 0's line of gain terms: free between the root and round 0)
        const NodeRef nd0 = node_ref(smem + L.offNodes, p.maxRow);
        // All the keys at once: in the graph whose nodes are the columns and whose arc j -> j' costs what column j pays for the
        // row that j' holds (its reduced cost, >= 0), the key of column c is the shortest cycle through c -- the diagonal of the
        // all-pairs closure with an empty diagonal to start from.  Floyd-Warshall in fp32 (the keys only ORDER the columns; any
        // order gives the same results), the matrix in the LDS of the node blocks and lists that round 0 has not touched yet:
        // D barriers and D^3 / threads min-plus steps instead of D searches of ~30 Dijkstra steps (64x64: 38 000 cycles
        // against 175 000).  Shapes whose matrix does not fit there run the searches.
        // (round 4: the closure is blocked, 12 barriers instead of 64 -- column_keys_closure, kbest_lap.h)
        const int fwOff = (L.offNodes + L.nodeStride + 15) & ~15;  // (16-byte aligned: the blocks are read four entries at a time)
        const int fwBytes = L.offRootMap - fwOff;
        const int Dp = (D + 15) & ~15;
        const bool fw = Dp * Dp * 4 <= fwBytes;
        if (fw) {
            column_keys_closure<NW>(reinterpret_cast<float *>(smem + fwOff), key, Cs, LDC, nd0.u, nd0.v, nd0.r4c, D, M);
        } else {
            const double v0 = (lane < D) ? nd0.v[lane] : 0.0;
            const int c4r0 = (lane < D) ? (int)nd0.c4r[lane] : -1;
            const int r4c0 = (lane < D) ? (int)nd0.r4c[lane] : -1;
#pragma unroll 1
            for (int c = wave; c < M; c += NW) {
                const int fr = __builtin_amdgcn_readlane(r4c0, c);
                const int c4r = (lane == fr) ? -1 : c4r0;
                double spc, delta;
                int pred, sink = 0;
                u64 scanned;
                const int st = dijkstra<false>(Cs, LDC, nd0.u, rl, lane, v0, c4r, allRows, bit64(fr), c, INF, spc, pred, scanned,
                                               delta, sink, 0.0, 0, parkFrom);
                if (lane == 0) key[c] = (st == 0) ? delta : INF;
            }
        }
        __syncthreads();
        if (wave == 0) {
            // positions by descending key, equal keys by column (deterministic)
            double kl = key[lane < M ? lane : 0];
            kl = (kl == kl) ? kl : INF;  // (a NaN key would break the ranks' uniqueness: any order is valid, a non-permutation is not)
            int rank = 0;
            for (int j = 0; j < M; j++) {
                const double kj = readlane_f64(kl, j);
                rank += (kj > kl || (kj == kl && j < lane)) ? 1 : 0;
            }
            if (lane < M) { posOf[lane] = (unsigned char)rank; colOf[rank] = (unsigned char)lane; }
            wave_fence();
            // the root in that order: u and row4col by position, col4row's values are positions, v as it is
            const int oc = (lane < M) ? (int)colOf[lane] : lane;
            const double uN = (lane < D) ? nd0.u[oc] : 0.0;
            const int r4cN = (lane < D) ? (int)nd0.r4c[oc] : -1;
            const int cOld = (lane < D) ? (int)nd0.c4r[lane] : -1;
            const int c4rN = (cOld >= 0 && cOld < M) ? (int)posOf[cOld] : cOld;
            const double vN = (lane < D) ? nd0.v[lane] : 0.0;
            const double g = nd0.gain[0];
            wave_fence();
            if (lane < D) nd0.u[lane] = uN;
            wave_fence();
            const u64 forb = bit64(__builtin_amdgcn_readlane(r4cN, 0));  // cpp:235: the row of the FIRST column of the order
            save_node(nd0, 0, vN, r4cN, c4rN, forb, g, 0);
            if (t0On && lane < D) rootMap[lane] = (unsigned char)c4rN;
        }
        __syncthreads();
        // the tile's real columns again, in the new order (the block is L2-resident)
        for (int c = wave; c < M; c += NW) {
            const int oc = colOf[c];
            for (int r = lane; r < N; r += 64) {
                const double x = Cg[r + (long long)oc * N];
                double val = noShift ? x : (maximize ? (-x + cdelTile) : (x - cdelTile));  // cpp:558 / cpp:564
                if (val != val) val = INF;
                Cs[r + c * LDC] = val;
            }
        }
        for (int i = tid; i < spec * 64; i += NT) { lbKey[i] = ~0ull; lbIn[i] = ~0u; }  // re-arm the filter minima (the key matrix lay there)
        __syncthreads();
    }

    // relay: the first piece hands over once k * relayFirst / 1024 solutions are out, the later ones relayStep / 1024 of k apart from there to
    // k (scalar integer arithmetic; the last piece and plain launches never: INT_MAX).  Kept in LDS and tested by the ONE lane
    // that counts the emitted solutions (phase D), which ends the round loop through ctrl->stop = 4: a test at the top of every
    // round by every wave (gridDim comes from memory) cost 2 % on every launch.
    if (RELAY) {
        if (tid == 0) {
            ctrl->relayCut = (piece + 1 < (int)gridDim.y)
                                 ? (int)(((long long)p.k * (p.relayFirst + p.relayStep * piece)) >> 10)
                                 : 0x7fffffff;
            if (fresh) ctrl->relayRound = 0;
            if (ctrl->stop == 4) ctrl->stop = 0;  // (a later piece: the image it took over ends with the hand-over code)
        }
        __syncthreads();
    }
    KB_ACC(0, __builtin_readcyclecounter() - profT0);  // [0] set-up + root solve
    // ---- phase 2: rounds ----------------------------------------------------------------------------
    int roundNo = RELAY ? uni32(ctrl->relayRound) : 0;  // (a later piece goes on with the round the image was taken before)
    for (; uni32(ctrl->stop) == 0; roundNo++) {
        // relay: this piece hands over once its share of the k solutions is out (the last piece runs to the end)
        // The compiler hoists every cheap expression of the lane number out of this loop (addresses, lane masks: two dozen of
        // them) and then has to spill them at 80 VGPRs: a scratch load per use instead of one or two vector instructions.
        // Making the lane number opaque once per round keeps those expressions where they are used.
        asm volatile("" : "+v"(lane), "+v"(tid));
        KB_T(tRound);
        KB_ACC(7, 1);  // [7] rounds
        if ((t0On && roundNo == 1) || (t1On && roundNo == 2)) {
            apriori_threshold<NW>(lbKey, atoms, &ctrl->t0, D, k, roundNo - 1, spec * 64);
#ifdef KB_T0_DEBUG
            __syncthreads();
            if (tid == 0 && b < 4) printf("T0DBG b=%d round=%d t0gap=%.6f poolT=%.6f nOld=%d extra=%d\n", b, roundNo, ctrl->t0 - __longlong_as_double((long long)atoms[2 * D]),
                                          (ctrl->nq - ctrl->head >= k - ctrl->emitted) ? PG[ctrl->head + k - ctrl->emitted - 1] - __longlong_as_double((long long)atoms[2 * D]) : -1.0, ctrl->nq - ctrl->head, (int)atoms[2 * D + 1]);
#endif
            // Every wave must have LEFT the function before its scratch is re-armed: the waves return one by one (behind the
            // function's last barrier each still reads the sorted atoms / the counters, through FLAT loads -- the function is
            // not inlined and takes generic pointers), and a wave that re-armed the minima under a slower one's reads gave
            // that wave another view of them.  Found in round 5 by the soak of 2-column frames at k = 1 025 (the first k
            // beyond the fused kernel once every launch enumerates k + 1): one problem in ten thousand came back with nf = 2;
            // with this barrier none in 560 000 (tests/test_gpu_round5.py::test_apriori_threshold_scratch_is_not_rearmed_early).
            __syncthreads();
            for (int i = tid; i < spec * 64; i += NT) lbKey[i] = ~0ull;  // re-arm the filter minima
            __syncthreads();
        }
        // control values come out of LDS in VGPRs: readfirstlane makes them provably wave-uniform, so every
        // loop below is scalar-controlled.  They are only rewritten in D, behind a barrier.
        const int nsel = uni32(ctrl->nsel);
        const int emitted = uni32(ctrl->emitted);
        const int R = k - emitted;  // candidates that can still be output
        const int nqEnd = uni32(ctrl->nq), head = uni32(ctrl->head);
        const int nOld = nqEnd - head;
        const int sidBase = uni32(ctrl->nextSid);
        const double cutG = ctrl->cutoffGain;
        // the candidates selected in the last A phase are split in this round: flag them in the pool now (nobody
        // reads the pool's meta words before the merge, which is behind the barrier after B)
        if (wave == 0 && lane < nsel) {
            const int idx = ctrl->selIdx[lane];
            if (idx >= 0) { PM[idx] |= META_SPLIT; PS[idx] = ctrl->selSid[lane]; }
        }
        // -- B1: first-step filter.  56 % of all children (64x64, k=200) are abandoned by the early-termination
        //    test at their very first Dijkstra step, i.e. because  min over candidate rows of (C[r,c] - u[c] - v[r])
        //    already exceeds the bound.  That minimum is computed here for ALL children of a node at once, one wave
        //    per node with lane = child column walking the rows (conflict-free thanks to the odd tile stride), and
        //    only the survivors are queued for a wave of their own.  It is the same test on the same numbers as
        //    step 1 of dijkstra<true> (delta = 0: (0 + C) - u - v), so the set of children that go on is unchanged.
        // threshold of the pool: once it holds R candidates only children below its largest can matter
        double T = (nOld >= R) ? PG[head + R - 1] : INF;
        if (useCut && !maximize && cutG < T) T = cutG;
        const double T0 = ctrl->t0;  // a-priori threshold (+inf until round 1, or when it is off)
#ifdef KB_T0_DEBUG
        if (tid == 0 && b < 2 && t0On) printf("T0DBG b=%d round=%02d poolgap=%.5f t0gap=%.5f emitted=%d nOld=%d\n", b, roundNo, T - __longlong_as_double((long long)atoms[2 * D]), T0 - __longlong_as_double((long long)atoms[2 * D]), emitted, nOld);
#endif
        if (T0 < T) T = T0;
        if (S > 1) {
            // The shares of one matrix exchange their thresholds: a share that holds k candidates at or below T has shown that the
            // GLOBAL k-th best gain is at most T, so every share may prune with the smallest T any of them has published
            // (atomicMin on the order-preserving key; a stale read only prunes later, never wrongly).
            u64 *sh = p.sharedT + b;
            if (wave == 0 && lane == 0 && T < INF) {
                int khi;
                u32 klo;
                to_key(T, khi, klo);
                atomicMin(sh, ((u64)((u32)khi ^ 0x80000000u) << 32) | klo);
            }
            const u64 kk = uni64(__hip_atomic_load(sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (kk != ~0ull) {
                const double Tsh = from_key((int)((u32)(kk >> 32) ^ 0x80000000u), (u32)kk);
                if (Tsh < T) T = Tsh;
                // (read again in A / D, behind two barriers: candidates beyond it are neither split nor emitted -- they cannot be
                //  among the global k best, and without this a share would go on to emit k hypotheses of its own)
                if (wave == 0 && lane == 0) ctrl->tShared = Tsh;
            }
        }
        const double cmaxv = ctrl->cmax;
        // (the optimistic bounds of this round's nodes were set when they were selected: opt->bAbs[w], struct Opt)
        const bool optAny = optOn && uni32(opt->anyFinite) != 0;
        KB_T(tF0);
        KB_ACC(14, tF0 - tRound);  // [14] round prologue (control reads)
        int myNode = -1, myParts = 1;  // the node whose filter this wave took part in
        {
            // all NW waves take part: node = wave % nsel, and the waves of one node split its columns j.
            // Walking the parent's columns j >= a instead of the rows makes the candidate test a scalar lane mask:
            // row r = row4col[j] is a candidate of child c iff c < j (rows of columns >= c, minus the row the child
            // frees itself, cpp:480-488 / 510-516); for the child on the active column (lane 0) it is a candidate
            // unless it is in the parent's accumulated forbidden set (cpp:490).
            // (nsel <= 8, NW <= 16: quotients by a 16-bit reciprocal, exact for these ranges -- an integer division
            //  costs ~25 scalar instructions, and every wave does three of them per round)
            const int rcpSel = RCP16[nsel];
            const int parts = (NW * rcpSel) >> 16;
            const int part = (wave * rcpSel) >> 16, nodeI = wave - part * nsel;
            if (part < parts) {
                myNode = nodeI;
                myParts = parts;
                const NodeRef nd = node_ref(smem + L.offNodes + (size_t)nodeI * L.nodeStride, p.maxRow);
                const int a = uni32(nd.info[0]);
                const u64 nforb = uni64(nd.forb[0]);
                const int c = a + lane;  // this lane's child column
                const int cc = c < D ? c : D - 1;
                const double uc = nd.u[cc];
                const double *Ccol = Cs + cc * LDC;
                const double vrow = (lane < D) ? nd.v[lane] : 0.0;        // lane = row
                const int r4cP = (lane < D) ? (int)nd.r4c[lane] : 0;      // lane = column
                const int span = D - a;
                const int rcpParts = RCP16[parts];
                const int jBeg = a + ((span * part * rcpParts) >> 16), jEnd = a + ((span * (part + 1) * rcpParts) >> 16);
                int mlo = 0, mhi = KEY_INF_HI;  // running minimum (+inf)
                // Column a itself contributes nothing: its row is in the forbidden set of the child on a, and no
                // other child keeps it.  From a + 1 on the mask of the lanes 1 .. (j - a) - 1 grows by one bit per
                // column (s_bitset1_b64), and lane 0 joins unless the row is forbidden for the active column.
                const u64 nallow = ~nforb;
                int j = jBeg > a ? jBeg : a + 1;
                u64 low = (j - a >= 64) ? ~1ull : (((1ull << (j - a)) - 1ull) & ~1ull);
                auto fetch = [&](int jj, double &cv, double &vr, u64 &mk) {
                    const int r = __builtin_amdgcn_readlane(r4cP, jj);
                    vr = readlane_f64(vrow, r);
                    cv = Ccol[r];
                    const u32 lo = (u32)low | ((u32)(nallow >> r) & 1u);
                    mk = (low & 0xffffffff00000000ull) | lo;
                    asm("s_bitset1_b64 %0, %1" : "+s"(low) : "s"(jj - a));
                };
                auto apply = [&](double cv, double vr, u64 mk) {
                    const double rc = (cv - uc) - vr;  // ((0 + C) - u) - v, cpp:313 with delta = 0
                    const u64 upd = __ballot(rc < __hiloint2double(mhi, mlo)) & mk;
                    mlo = sel32(upd, __double2loint(rc), mlo);
                    mhi = sel32(upd, __double2hiint(rc), mhi);
                };
                for (; j + 4 <= jEnd; j += 4) {  // four independent LDS reads in flight
                    double cv[4], vr[4];
                    u64 lm[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) fetch(j + i, cv[i], vr[i], lm[i]);
#pragma unroll
                    for (int i = 0; i < 4; i++) apply(cv[i], vr[i], lm[i]);
                }
                for (; j < jEnd; j++) {
                    double cv, vr;
                    u64 mk;
                    fetch(j, cv, vr, mk);
                    apply(cv, vr, mk);
                }
                const double m = __hiloint2double(mhi, mlo);
                // combine the row parts: integer min of the order-preserving key
                int khi;
                u32 klo;
                to_key(m, khi, klo);
                const u64 key = ((u64)((u32)khi ^ 0x80000000u) << 32) | klo;  // signed high word -> unsigned order
                if (c < M) atomicMin(&lbKey[nodeI * 64 + lane], key);
                // Second pass, the other end of the path: the only unassigned row of child c is the row it frees,
                // fr = row4col[c], so its path ENDS with an arc (fr, j), j a later column.  minIn[c] = min over j > c
                // of (C[fr,j] - u[j]) - v[fr] (high word, clamped at 0: a lower bound) -- first arc and last arc are
                // different arcs of the same path and all reduced costs are >= 0, so first + last > bound kills the
                // child here, and what survives starts its search against bound - minIn.
                if (prune && (T < INF || optAny)) {
                    const int fr = (int)nd.r4c[cc];
                    const double vfr = nd.v[fr];
                    const double *Crow = Cs + fr;
                    int inHi = KEY_INF_HI;
                    int j2 = jBeg > a ? jBeg : a + 1;
                    u64 mk = (j2 - a >= 64) ? ~0ull : ((1ull << (j2 - a)) - 1ull);  // lanes 0 .. (j - a) - 1: children before column j
                    auto last_arc = [&](int jj, double cin, double uj) {
                        const double rin = (cin - uj) - vfr;
                        int h = __double2hiint(rin);
                        h = h < 0 ? 0 : h;  // -1e-17 from rounding: no information
                        h = sel32(mk, h, KEY_INF_HI);
                        inHi = h < inHi ? h : inHi;
                        asm("s_bitset1_b64 %0, %1" : "+s"(mk) : "s"(jj - a));
                    };
                    for (; j2 + 4 <= jEnd; j2 += 4) {  // four independent pairs of LDS reads in flight
                        double cin[4], uj[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) { cin[i] = Crow[(j2 + i) * LDC]; uj[i] = nd.u[j2 + i]; }
#pragma unroll
                        for (int i = 0; i < 4; i++) last_arc(j2 + i, cin[i], uj[i]);
                    }
                    for (; j2 < jEnd; j2++) last_arc(j2, Crow[j2 * LDC], nd.u[j2]);
                    if (c < M) atomicMin(&lbIn[nodeI * 64 + lane], (u32)inHi);
                }
            }
        }
        KB_T(tF1);
        KB_ACC(15, tF1 - tF0);     // [15] first-step filter busy
        // The wave that finishes a node's filter LAST compacts its survivors (no barrier between the two): the minima of
        // all parts are in LDS once every part has bumped the node's counter (a wave's LDS operations execute in order).
        bool compactor = false;
        if (myNode >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            int old = 0;
            if (lane == 0) old = atomicAdd(&ctrl->partsDone[myNode], 1);
            compactor = uni32(old) == myParts - 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (compactor) {  // one wave per node: survivors = finite minimum (else infeasible, cpp:327) within the bound
            if (lane == 0) ctrl->partsDone[myNode] = 0;
            const NodeRef nd = node_ref(smem + L.offNodes + (size_t)myNode * L.nodeStride, p.maxRow);
            const int a = uni32(nd.info[0]);
            const int sid = uni32(nd.info[1]);
            const double pgain = nd.gain[0];
            // boundV: against the pool's valid threshold; bound: against this node's optimistic bound (<= boundV).  A child between
            // the two is deferred: its lower bound goes into the node's ticket (struct Opt)
            const double boundV = (prune && T < INF) ? (T - pgain) + 1e-9 * (fabs(T) + cmaxv) : INF;
            double bAbs = T;
            if (optOn) { const double bo = opt->bAbs[myNode]; bAbs = bo < T ? bo : T; }
            const double bound = (prune && bAbs < INF) ? (bAbs - pgain) + 1e-9 * (fabs(bAbs) + cmaxv) : INF;
            const u64 doneM = optOn ? uni64(opt->done[myNode]) : 0ull;  // children completed in an earlier split of this node
            const int c = a + lane;
            // (root-subtree sharding partitions on the REFERENCE's column, kbest_c.h: the enumeration's own order depends on the launch
            //  shape and the knobs, and shards that differ in them must still enumerate disjoint, complete partitions)
            bool live = c < M && !((doneM >> (c & 63)) & 1ull);
            if (__builtin_expect(sid == 0 && rcStride > 1, 0)) {
                // (the stride is made opaque here: the compiler would otherwise keep the reciprocal of this division in a register
                //  across the whole round loop -- and spill it -- for a test that only the root of a sharded run ever makes)
                int st = rcStride;
                asm volatile("" : "+s"(st));
                live = live && ((int)colOf[c & 63] % st) == rcOffset;
            }
            const u64 key = lbKey[myNode * 64 + lane];
            const double m = from_key((int)((u32)(key >> 32) ^ 0x80000000u), (u32)key);
            const u32 inH = lbIn[myNode * 64 + lane];
            const double minIn = (prune && (T < INF || optAny)) ? __hiloint2double((int)inH, 0) : 0.0;  // +inf: no last arc at all
            const bool keep = live && m < INF && !(m + minIn > bound);
            if (optOn && live && (m + minIn) < INF && (m + minIn > bound) && !(m + minIn > boundV)) {
                // deferred at the filter: first arc + last arc is a lower bound of the child's distance
                int khi;
                u32 klo;
                to_key(pgain + (m + minIn), khi, klo);
                atomicMin(&opt->defKey[myNode], ((u64)((u32)khi ^ 0x80000000u) << 32) | klo);
            }
            // Children whose first step is far inside the bound tend to run long (they are the ones that complete):
            // they are queued from the front, the others from the back, so that the long ones start first and the
            // round does not end waiting for one late straggler.
            const bool heavy = keep && !(m > 0.5 * bound);
            const u64 kh = __ballot(heavy), kl = __ballot(keep && !heavy);
            int baseH = 0, baseL = 0;
            if (lane == 0 && kh) baseH = atomicAdd(&ctrl->nSurv, __popcll(kh));
            if (lane == 0 && kl) baseL = atomicAdd(&ctrl->nSurvBack, __popcll(kl));
            baseH = uni32(baseH);
            baseL = uni32(baseL);
            const u64 below = (1ull << lane) - 1ull;
            // survivor entry (16 bits): column (6), node (4) and the last-arc bound as a fraction of the node's bound in
            // 63ths, rounded down (6 bits): what survived has minIn <= bound
            int q = 0;
            if (keep && bound < INF && bound > 0.0) {
                const double f = (minIn / bound) * 63.0;
                q = f >= 63.0 ? 63 : (int)f;
            }
            const unsigned short entry = (unsigned short)((q << 10) | (myNode << 6) | c);
            if (heavy) surv[baseH + __popcll(kh & below)] = entry;
            else if (keep) surv[spec * 64 - 1 - (baseL + __popcll(kl & below))] = entry;
        }
        __syncthreads();
        // -- outputs, as they become final: the slots emitted up to the last round (their states are saved, their slot ->
        //    state entries written behind a barrier) are widened into row4col / col4row NOW, one slot per wave.  The result tables then leave the kernel spread over its
        //    whole run instead of in one burst at the end of every matrix -- which is what a host that takes them over PCIe
        //    (result tables in pinned host memory, kbest_batch_f64) would otherwise wait for after the last matrix.  At the
        //    START of the children phase: the stores (microseconds each when they cross PCIe) drain while the children are
        //    solved, not in front of a barrier.
        {
            const int outDone = uni32(ctrl->outDone);
            for (;;) {
                int t = 0;
                if (lane == 0) t = atomicAdd(&ctrl->outTicket, 1);
                const int sOut = outDone + uni32(t);
                if (sOut >= emitted) break;
                if (sOut >= kTab) continue;  // (the solution behind the tables: its gain is all that is kept)
                const unsigned char *st = stBase + (long long)slotSid[sOut] * p.stateStride;
                // (the states are in the enumeration's column order: the tables in the reference's)
                if (lane < M) put_index(p.row4col, (outBase + sOut) * p.ldCol + colOf[lane], st[offR4C + lane], tabI8);
                if (p.col4row && lane < N) {
                    const int cv = st[offC4R + lane];
                    put_index(p.col4row, (outBase + sOut) * p.ldRow + lane, (rect && cv == 255) ? -1 : (cv < M ? (int)colOf[cv] : cv), tabI8);  // unassigned row (cpp:134)
                }
            }
        }
        // -- B2: surviving children (shortestPathUpdateCPP, gain only), dynamic queue over the survivor list.  The
        //    per-node data a wave needs is cached in registers across consecutive items of the same node, and the
        //    next queue ticket is drawn before the current child is solved so that its LDS round trip is hidden.
        const int nFront = uni32(ctrl->nSurv);
        const int totalItems = nFront + uni32(ctrl->nSurvBack);
        {
            int npush = 0;
            int curW = -1, a = 0, sid = 0;
            double v = 0.0, bound = INF;
            int c4rP = -1, r4cP = -1;
            u64 nforb = 0;
            NodeRef nd = node_ref(smem + L.offNodes, p.maxRow);
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&ctrl->nextItem, 1);
            for (;;) {
                KB_T(tItem);
                const int item = uni32(ticket);
                if (item >= totalItems) break;
                if (lane == 0) ticket = atomicAdd(&ctrl->nextItem, 1);  // prefetch the next ticket
                KB_ACC(4, 1);  // [4] children started
                const int sv = uni32((int)surv[item < nFront ? item : spec * 64 - 1 - (item - nFront)]);
                const int w = (sv >> 6) & 15, c = sv & 63;
                if (w != curW) {  // (re)load this node's data
                    curW = w;
                    nd = node_ref(smem + L.offNodes + (size_t)w * L.nodeStride, p.maxRow);
                    a = uni32(nd.info[0]);
                    sid = uni32(nd.info[1]);
                    const double pgain = nd.gain[0];
                    nforb = uni64(nd.forb[0]);
                    v = (lane < D) ? nd.v[lane] : 0.0;
                    c4rP = (lane < D) ? (int)nd.c4r[lane] : -1;
                    r4cP = (lane < D) ? (int)nd.r4c[lane] : -1;
                    // early-termination bound on the Dijkstra distance: child gain = parent gain + delta (up to
                    // rounding), so delta > (T - parent gain) + margin can never enter the k best.  The search runs against the
                    // node's optimistic bound (struct Opt); boundV is the valid one.
                    double bAbsN = T;
                    if (optOn) { const double bo = opt->bAbs[w]; bAbsN = bo < T ? bo : T; }
                    bound = (prune && bAbsN < INF) ? (bAbsN - pgain) + 1e-9 * (fabs(bAbsN) + cmaxv) : INF;
                }
                const int fr = __builtin_amdgcn_readlane(r4cP, c);           // row freed: cpp:277-278
                const u64 cand = __ballot(lane < D && c4rP >= c);             // rows of columns >= c: cpp:480-488, 525-527
                const u64 forbm = (c == a) ? nforb : bit64(fr);                // cpp:490 / cpp:510-516
                const int c4r = (lane == fr) ? -1 : c4rP;
                double spc, delta;
                int pred, sink = 0;
                u64 scanned;
                // backward bound from the filter (in 63ths of the bound, rounded down): the loop runs against bound - minIn
                const double minIn = (bound < INF) ? (double)(sv >> 10) * (bound * (1.0 / 63.0)) : 0.0;
                KB_T(tDij0);
                KB_ACC(9, tDij0 - tItem);  // [9] per-child set-up cycles
                const int st = dijkstra<true>(Cs, LDC, nd.u, rl, lane, v, c4r, cand, forbm, c, bound, spc, pred,
                                              scanned, delta, sink, minIn, fr, parkFrom);
                KB_T(tDij1);
                KB_ACC(8, tDij1 - tDij0);  // [8] cycles inside child Dijkstra
                KB_ACC(5, __popcll(scanned) + (st != 0));  // [5] child Dijkstra steps (approx: scanned rows)
                if (st != 0) {
                    // given up against the optimistic bound: the child's gain is beyond that bound, and at least parent gain +
                    // the distance settled so far -- deferred (the node's ticket) unless that is beyond the valid bound too
                    // (the node's bounds are fetched again here rather than kept in registers across the child loop)
                    if (optOn && st == 2) {
                        const double pg = nd.gain[0], bo = opt->bAbs[w];  // (two independent LDS reads)
                        const double boundV = (T - pg) + 1e-9 * (fabs(T) + cmaxv);  // (+inf with T)
                        if (bo < T && !(delta > boundV) && lane == 0) {
                            const double lbAbs = pg + delta;
                            int khi;
                            u32 klo;
                            to_key(lbAbs > bo ? lbAbs : bo, khi, klo);
                            atomicMin(&opt->defKey[w], ((u64)((u32)khi ^ 0x80000000u) << 32) | klo);
                        }
                    }
                    continue;
                }
                if (optOn && lane == 0) atomicOr(&opt->done[w], 1ull << c);  // never generated again when the node is split again
                KB_ACC(6, 1);  // [6] children completed
                // The child survived: finish it the way shortestPathUpdateCPP does -- path flip (cpp:108-116), exact
                // gain (calcGain, cpp:59-80) and,
                // if a state slot is free, the dual update (cpp:92-106) -- and keep the whole hypothesis, so that it
                // can later be split without being solved again.  Without a slot it stays a lazy candidate.
                int r4c = (lane == c) ? -1 : r4cP;
                int c4rN = c4r;
                {
                    int r = sink, cc, guard = 0;
                    do {
                        cc = __builtin_amdgcn_readlane(pred, r);
                        const int nxt = __builtin_amdgcn_readlane(r4c, cc);
                        c4rN = (lane == r) ? cc : c4rN;
                        r4c = (lane == cc) ? r : r4c;
                        r = nxt;
                    } while (cc != c && ++guard < 64);
                }
                const double g = serial_gain(Cs, LDC, lane, r4c, M, gainW, colOf);
                if (useCut && (maximize ? (g < cutG) : (g > cutG))) continue;  // cutHyp, cpp:496/521
                npush++;
                if (t1On && roundNo == 1) {
                    // a child of one of the root's best children: if it differs from the OPTIMUM by one cycle too (it mostly
                    // is another way through the region its parent's path covers) it is one more atom for the second
                    // threshold.  Rows moved against the optimum; one cycle <=> the orbit of a moved row under
                    // "who holds my old column now" covers them all.
                    const int rootC = (lane < D) ? (int)rootMap[lane] : -1;
                    const u64 moved = __ballot(lane < D && c4rN != rootC);
                    u64 seen = 0;
                    if (moved) {
                        const int r0 = __builtin_ctzll(moved);
                        int r = r0, guard = 0;
                        do {
                            seen |= 1ull << r;
                            const int col = __builtin_amdgcn_readlane(rootC, r);
                            r = __builtin_amdgcn_readlane(r4c, col);
                        } while (r != r0 && ++guard < 64);
                    }
                    if (moved != 0ull && seen == moved && lane == 0) {
                        const unsigned pos = atomicAdd(reinterpret_cast<unsigned *>(&atoms[2 * D + 1]), 1u);
                        if (pos < (unsigned)T0_EXTRA) {
                            atoms[2 * D + 2 + 2 * pos] = (u64)__double_as_longlong(g - __longlong_as_double((long long)atoms[2 * D]));
                            atoms[2 * D + 3 + 2 * pos] = moved;
                        }
                    }
                }
                if (t0On && sid == 0) {
                    // a child of the root: the optimum with ONE alternating path/cycle applied.  Rows it moves (rows on
                    // zero-padded columns count as one place):
                    const int cOld = c4rP >= M ? M : c4rP, cNew = c4rN >= M ? M : c4rN;
                    const u64 moved = __ballot(lane < D && cOld != cNew);
                    if (lane == 0) {
                        atoms[2 * c] = (u64)__double_as_longlong(g - nd.gain[0]);
                        atoms[2 * c + 1] = moved;
                    }
                }
                int slot = -1;
                if (lane == 0) {
                    const int sl = atomicAdd(&ctrl->nextEager, 1);
                    slot = sl < maxSid ? sl : -1;
                }
                slot = uni32(slot);
                if (slot >= 0) {
                    const bool sc = ((scanned >> lane) & 1ull) != 0;
                    const double vN = sc ? (v - delta + spc) : v;                      // cpp:102-106
                    const int rowOfCol = (lane < D && lane != c) ? r4cP : 0;            // parent's row of this column
                    const double spcOfRow = __hiloint2double(__shfl(__double2hiint(spc), rowOfCol),
                                                             __shfl(__double2loint(spc), rowOfCol));
                    double uN = (lane < D) ? nd.u[lane] : 0.0;
                    if (lane < D && lane != c && ((scanned >> rowOfCol) & 1ull)) uN = uN + delta - spcOfRow;  // cpp:96-99
                    if (lane == c) uN = uN + delta;                                     // cpp:92
                    const u64 forbN = forbm | bit64(__builtin_amdgcn_readlane(r4c, c));  // cpp:362
                    store_state(slot, uN, vN, r4c, c4rN, forbN, g, c);
                }
                KB_ACC(10, __builtin_readcyclecounter() - tDij1);  // [10] finish of completed children
                if (lane == 0) {
                    const int pos = atomicAdd(&ctrl->nFresh, 1);
                    freshG[pos] = g;
                    freshM[pos] = ((u32)sid << 8) | (u32)c;  // (parent state, column) of this candidate
                    freshS[pos] = slot >= 0 ? (unsigned short)slot : SID_NONE;
                }
            }
            if ((p.flags & KBEST_FLAG_COUNT_PUSHED) && lane == 0 && npush) atomicAdd(&ctrl->pushed, npush);
        }
        KB_T(tB1);
        KB_ACC(1, tB1 - tRound);  // [1] phase B busy (this wave)
        __syncthreads();
        KB_T(tC0);
        KB_ACC(11, tC0 - tB1);    // [11] wait at the barrier after B
        // -- C: rank-merge the fresh candidates into the sorted pool IN PLACE (every thread first pulls its
        //    entries into registers), keep the R smallest.  Ties in gain are ordered by (parent, column), so the
        //    result does not depend on the arrival order of the fresh list.
        const int nFresh = uni32(ctrl->nFresh);
        if (optOn && wave == NW - 1) {
            // The tickets (struct Opt), one wave, lane = node for the new ones and lane = list index for the old ones: the
            // nodes split in this round get their `done` masks saved with their states and, where children were deferred, a
            // ticket keyed by the smallest lower bound among them (minus twice the pruning margin: the bounds are sums of
            // rounded terms, the gains they are compared with are serial sums); tickets consumed by this round's re-splits
            // and tickets beyond the valid threshold leave the list; the list stays sorted.
            const int nTo = uni32(opt->nT);
            const u32 selTicket = (u32)uni32((int)opt->selTicket);
            const bool isNode = lane < nsel;
            const u64 dk = isNode ? opt->defKey[lane] : ~0ull;
            if (__ballot(dk != ~0ull) != 0ull || selTicket != 0u || nTo != 0) {
            const int sidL = isNode ? (int)ctrl->selSid[lane] : 0;
            double nK = INF;
            if (dk != ~0ull) {
                const double lb = from_key((int)((u32)(dk >> 32) ^ 0x80000000u), (u32)dk);
                nK = lb - 2e-9 * (fabs(lb) + cmaxv);
            }
            const bool hasNew = isNode && dk != ~0ull && !(nK > T);  // (beyond the valid threshold: nothing of it can be output)
            if (hasNew) *reinterpret_cast<u64 *>(stBase + (long long)sidL * p.stateStride + offDone) = opt->done[lane];
            if (isNode) opt->defKey[lane] = ~0ull;
            const double oK = lane < nTo ? opt->TK[lane] : INF;
            const int oS = lane < nTo ? (int)opt->TS[lane] : -1;
            const bool consumed = lane < (int)selTicket;  // (selected in list order, and the list has not changed since)
            const bool oKeep = lane < nTo && !consumed && !(oK > T);
            const u64 keepM = __ballot(oKeep);
            u64 newM = __ballot(hasNew);
            int posO = __popcll(keepM & ((1ull << lane) - 1ull)), posN = 0;
            for (u64 mm = newM; mm; mm &= mm - 1) {
                const int w = __builtin_ctzll(mm);
                const double kw = readlane_f64(nK, w);
                posO += (kw < oK) ? 1 : 0;                                    // old before new among equal keys
                const int cntOld = __popcll(__ballot(oKeep && oK <= kw));
                posN = (lane == w) ? posN + cntOld : posN;
                posN += (hasNew && lane != w && (kw < nK || (kw == nK && w < lane))) ? 1 : 0;
            }
            wave_fence();
            if (oKeep) { opt->TK[posO] = oK; opt->TS[posO] = (unsigned short)oS; }
            if (hasNew) { opt->TK[posN] = nK; opt->TS[posN] = (unsigned short)sidL; }
            if (lane == 0) opt->nT = __popcll(keepM) + __popcll(newM);
            }
        }
        double og[EPT];
        u32 om[EPT];
        unsigned short os[EPT];
        int opos[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * NT;
            opos[e] = -1;
            if (i < nOld) {
                const double g = PG[head + i];
                int pos = i;
                {   // four fresh gains per pair of 16-byte broadcast reads: the loop is latency, not arithmetic
                    const double2 *f2 = reinterpret_cast<const double2 *>(freshG);
                    int j = 0;
                    for (; j + 4 <= nFresh; j += 4) {
                        const double2 a = f2[j >> 1], b = f2[(j >> 1) + 1];
                        pos += ((a.x < g) ? 1 : 0) + ((a.y < g) ? 1 : 0) + ((b.x < g) ? 1 : 0) + ((b.y < g) ? 1 : 0);
                    }
                    for (; j < nFresh; j++) pos += (freshG[j] < g) ? 1 : 0;
                }
                og[e] = g;
                om[e] = PM[head + i];
                os[e] = PS[head + i];
                opos[e] = pos;
            }
        }
        double fg = 0.0;
        u32 fm = 0;
        unsigned short fs = SID_NONE;
        int fpos = -1;
        if (tid < nFresh) {  // nFresh <= spec * 64 <= NT
            const double g = freshG[tid];
            const u32 mj = freshM[tid];
            int lo = 0, hi = nOld;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (PG[head + mid] <= g) lo = mid + 1; else hi = mid;
            }
            int pos = lo;
            {
                const double2 *f2 = reinterpret_cast<const double2 *>(freshG);
                int nEq = 0, j2 = 0;
                for (; j2 + 4 <= nFresh; j2 += 4) {
                    const double2 a = f2[j2 >> 1], b = f2[(j2 >> 1) + 1];
                    pos += ((a.x < g) ? 1 : 0) + ((a.y < g) ? 1 : 0) + ((b.x < g) ? 1 : 0) + ((b.y < g) ? 1 : 0);
                    nEq += ((a.x == g) ? 1 : 0) + ((a.y == g) ? 1 : 0) + ((b.x == g) ? 1 : 0) + ((b.y == g) ? 1 : 0);
                }
                for (; j2 < nFresh; j2++) {
                    const double g2 = freshG[j2];
                    pos += (g2 < g) ? 1 : 0;
                    nEq += (g2 == g) ? 1 : 0;
                }
                if (__builtin_expect(nEq > 1, 0))  // another fresh candidate with the same gain: order by (parent, column)
                    for (int j3 = 0; j3 < nFresh; j3++) pos += (freshG[j3] == g && freshM[j3] < mj) ? 1 : 0;
            }
            fg = g; fm = mj; fs = freshS[tid]; fpos = pos;
        }
        __syncthreads();
        for (int i = tid; i < spec * 64; i += NT) { lbKey[i] = ~0ull; lbIn[i] = ~0u; }  // the fresh lists are consumed: re-arm the minima
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (opos[e] >= 0 && opos[e] < R) { PG[opos[e]] = og[e]; PM[opos[e]] = om[e]; PS[opos[e]] = os[e]; }
        if (fpos >= 0 && fpos < R) { PG[fpos] = fg; PM[fpos] = fm; PS[fpos] = fs; }
        int nq = nOld + nFresh;
        if (nq > R) nq = R;
        __syncthreads();
        KB_T(tA0);
        KB_ACC(2, tA0 - tC0);     // [2] merge incl. its barrier
        // -- A + D: every wave finds the candidates to split next itself (the pool is stable now and nothing
        //    below writes to it): the first `budget` entries that have not been split yet.  Wave w brings the
        //    w-th of them into node block w -- by loading its saved state, or, if it has none, by re-solving it
        //    from its parent's state.  Lane 0 of wave 0 also does the emission bookkeeping, which depends only on
        //    the pool order and flags.  Lazy state slots: keep one in hand for every output still to come.
        const double tShared = (S > 1) ? ctrl->tShared : INF;
        int budget = (p.lazyStates - k) + 1 - (sidBase - emitted);
        if (budget > spec) budget = spec;
        if (budget < 1) budget = 1;
        int mySel = -1, mySid = 0, nselNew = 0, nLazy = 0;
        constexpr int MS = (NW >= 16) ? 16 : (NW >= 12 ? 12 : 8);  // candidates split per round at most
        int sIdx[MS], sSid[MS];
#pragma unroll
        for (int w = 0; w < MS; w++) { sIdx[w] = -1; sSid[w] = 0; }
        // wave 0 walks the whole selection (it writes the control block); wave w only as far as its own, the w-th
        const int walk = (wave == 0 || budget <= wave) ? budget : wave + 1;
        int firstOpen = -1;  // pool index of the first candidate that has not been split
        // the candidates: selection ranks rank0, rank0 + 1, ... go to the first open entries of the pool, in pool order
        auto walk_candidates = [&](int rank0) {
            nselNew = rank0;
            nLazy = 0;
            for (int base = 0; base < nq && nselNew < walk; base += 64) {
                const int i = base + lane;
                const bool open = i < nq && !(PM[i] & META_SPLIT) && !(S > 1 && PG[i] > tShared);
                const unsigned short ps = (i < nq) ? PS[i] : SID_NONE;
                u64 m = __ballot(open);
                const u64 lazyM = __ballot(open && ps == SID_NONE);
                if (m && firstOpen < 0) firstOpen = base + __builtin_ctzll(m);
                while (m && nselNew < walk) {
                    const int bitpos = __builtin_ctzll(m);
                    const bool lazy = (lazyM >> bitpos) & 1ull;
                    const int sidv = lazy ? sidBase + nLazy : __builtin_amdgcn_readlane((int)ps, bitpos);
                    if (nselNew == wave) { mySel = base + bitpos; mySid = sidv; }
                    if (wave == 0) {
#pragma unroll
                        for (int w = 0; w < MS; w++) if (w == nselNew) { sIdx[w] = base + bitpos; sSid[w] = sidv; }
                    }
                    nLazy += lazy ? 1 : 0;
                    nselNew++;
                    m &= m - 1;
                }
            }
        };
        walk_candidates(0);
        // Re-split tickets (struct Opt): a ticket is DUE when its key is at or below the gain of the first open candidate (it
        // would block the emission of everything from there on); the due tickets -- the first tSel of the sorted list -- take the
        // first selection ranks of the round and the candidates move up.  Rare (a few per matrix): a second walk then.
        const int nTl = optOn ? uni32(opt->nT) : 0;
        int tSel = 0;
        if (nTl > 0) {
            const double g1 = firstOpen >= 0 ? PG[firstOpen] : INF;
            const double tk = lane < nTl ? opt->TK[lane] : INF;
            tSel = __popcll(__ballot(lane < nTl && tk <= g1));
            tSel = tSel > OPT_TSEL ? OPT_TSEL : tSel;
            tSel = tSel > budget ? budget : tSel;
            if (tSel > 0) {
                mySel = -1;
#pragma unroll
                for (int w = 0; w < MS; w++) { sIdx[w] = -1; sSid[w] = 0; }
                walk_candidates(tSel);
                if (wave < tSel) { mySel = -1; mySid = uni32((int)opt->TS[wave]); }
                if (wave == 0) {
#pragma unroll
                    for (int w = 0; w < MS; w++) if (w < tSel) { sIdx[w] = -1; sSid[w] = uni32((int)opt->TS[w]); }
                }
            }
        }
        const bool myTicket = wave < tSel;
        if (optOn && wave == NW - 1) {
            // The optimistic bound of the nodes just selected (struct Opt), by the LAST wave -- it has walked the whole selection, and
            // wave 0, whose emission bookkeeping is the critical path of this phase, is left alone --: the gain at the rho-quantile of
            // the pool's candidates, rho growing from optRho0 to optRho1 while the first optPhi * k solutions go out (early on most
            // of the pool is speculation, late most of it is the answer); for a ticket at least a step beyond its key (progress).
            // ANY value is correct -- the users take the minimum with the round's valid threshold --, so the pool is taken as it
            // stands before this round's emission; with no room for the round's tickets there is no guess.
            double bA = INF;
            if (nq >= p.optMinPool && nTl + nselNew <= OPT_TICKETS) {
                const int n = nq < R ? nq : R;
                float rho = p.optRho0 + p.optSlope * (float)emitted;  // (optSlope: kbest_engine.h)
                rho = (p.optSlope >= 0.0f) == (rho > p.optRho1) ? p.optRho1 : rho;
                int qi = (int)(rho * (float)n);
                qi = qi >= n ? n - 1 : (qi < 0 ? 0 : qi);
                bA = PG[qi];
            }
            if (lane < tSel) {  // (selection rank = place in the list)
                const double K = opt->TK[lane];
                const double up = K + p.optKappa * (K - opt->gRoot) + 8e-9 * (fabs(K) + cmaxv);
                bA = up > bA ? up : bA;
            }
            if (lane < nselNew) opt->bAbs[lane] = bA;
            const u64 fin = __ballot(lane < nselNew && bA < INF);
            if (lane == 0) { opt->anyFinite = fin != 0ull ? 1 : 0; opt->selTicket = (u32)tSel; }
        }
        // The saved state of this wave's node: the loads are issued here and land in registers while wave 0 does the
        // emission bookkeeping below (it has a node of its own to bring in, and would otherwise be the last at the barrier
        // every round by exactly that bookkeeping).
        const bool haveNode = wave < nselNew;
        const bool lazyNode = haveNode && !myTicket && uni32((int)PS[mySel < 0 ? 0 : mySel]) == (int)SID_NONE;
        double ldU = 0.0, ldV = 0.0, ldGain = 0.0;
        int ldR = 0, ldC = 0, ldA = 0;
        u64 ldForb = 0;
        if (haveNode && !lazyNode) {
            const unsigned char *st = stBase + (long long)mySid * p.stateStride;
            const double *sd = reinterpret_cast<const double *>(st);
            if (lane < D) {
                ldU = sd[lane];
                ldV = sd[DS + lane];
                ldR = st[offR4C + lane];
                ldC = st[offC4R + lane];
            }
            if (lane == 0) {
                ldForb = *reinterpret_cast<const u64 *>(st + offTail);
                ldGain = *reinterpret_cast<const double *>(st + offTail + 8);
                ldA = *reinterpret_cast<const int *>(st + offTail + 16);
            }
        }
        if (wave == 0) {
            // emission (kBest2D cpp:607-634), all lanes of wave 0, 64 pool entries per pass: the head goes out while it has
            // been split; the first entry that has NOT been split is the first one selected in this round (selection is in
            // pool order): it is emitted too but ends the run, because its children are not in the pool yet.  (One lane walking
            // the entries one by one -- five dependent LDS reads each -- kept the other eleven waves at the barrier below for
            // ~10 000 cycles per round.)
            int e = emitted, h = 0, stop = 0;
            const double cdel = ctrl->cdelta, g0u = ctrl->gain0u;
            const double tMin = nTl > 0 ? opt->TK[0] : INF;  // emission stops at the smallest ticket key (struct Opt)
            int sidFirst = sSid[0];                          // state slot of the first selected CANDIDATE (rank tSel)
#pragma unroll
            for (int w = 1; w < MS; w++) sidFirst = (w == tSel) ? sSid[w] : sidFirst;
            bool more = true;
            for (int base = 0; more && base < nq && e < k; base += 64) {
                const int i = base + lane;
                const bool valid = i < nq;
                const double g = valid ? PG[i] : 0.0;
                const u32 meta = valid ? PM[i] : 0u;
                const int psid = valid ? (int)PS[i] : 0;
                const double gu = maximize ? (-g + cdel) : (g + cdel);  // cpp:626-630
                const bool cutB = useCut && valid && (maximize ? (gu < g0u - p.cutoff) : (gu > g0u + p.cutoff));
                const bool shB = S > 1 && valid && g > tShared;  // beyond the global k-th best: this share is done
                const bool tickB = valid && !(g < tMin);  // a re-split ticket comes first: its node may still hold something better
                const u64 plainM = __ballot(valid && (meta & META_SPLIT) && !cutB && !shB && !tickB);
                int run = (~plainM == 0ull) ? 64 : __builtin_ctzll(~plainM);  // leading entries that simply go out
                if (run > k - e) run = k - e;
                if (lane < run) {
                    if (e + lane < kTab) p.gain[outBase + e + lane] = gu;
                    else p.tieGain[blk] = gu;  // (tie mode only: the solution behind the tables)
                    slotSid[e + lane] = (unsigned short)psid;
                }
                e += run;
                h += run;
                if (run == 64) continue;     // the whole pass went out: next 64 entries
                more = false;
                if (e >= k || base + run >= nq) break;
                // the entry that ended the run
                const bool tSh = (__ballot(shB) >> run) & 1ull, tSplit = (__ballot((meta & META_SPLIT) != 0) >> run) & 1ull;
                const bool tCut = (__ballot(cutB) >> run) & 1ull;
                if (tSh) { stop = 1; break; }
                if ((__ballot(tickB) >> run) & 1ull) break;  // behind a ticket: wait for its node's re-split
                if (!tSplit && nselNew <= tSel) break;  // not split and not selected this round: wait
                if (lane == run) {
                    if (e < kTab) p.gain[outBase + e] = gu;
                    else if (!tCut) p.tieGain[blk] = gu;  // (beyond the cutoff: written in the reference, never counted)
                    slotSid[e] = (unsigned short)(tSplit ? psid : sidFirst);
                }
                if (tCut) { stop = 1; break; }  // cpp:709-719: slot written, not counted
                e++;
                h++;
                // (a split entry beyond the cutoff cannot get here; a fresh one ends the run)
            }
            if (e >= k) stop = 1;
            if (h >= nq && nselNew == 0) stop = 1;  // queue empty, nothing left to split: cpp:631-633
            if (lane == 0) {
                ctrl->emitted = e;
                ctrl->nsel = nselNew;
                ctrl->nextSid = sidBase + nLazy;
                ctrl->nextItem = 0;
                ctrl->nFresh = 0;
                ctrl->nSurv = 0;
                ctrl->nSurvBack = 0;
                ctrl->outDone = emitted;  // (the slots emitted before this round: written above, before the barrier after B)
                ctrl->outTicket = 0;
                ctrl->nq = nq;
                ctrl->head = h;
#pragma unroll
                for (int w = 0; w < MS; w++) { ctrl->selIdx[w] = (short)sIdx[w]; ctrl->selSid[w] = (unsigned short)sSid[w]; }
                if (stop) ctrl->stop = 1;
                else if (RELAY && e >= ctrl->relayCut) ctrl->stop = 4;  // relay: this piece's share is out -- the round ends as usual, the loop with it
            }
        }
        if (haveNode) {
            const NodeRef nd = node_ref(smem + L.offNodes + (size_t)wave * L.nodeStride, p.maxRow);
            // children completed in an earlier split of this node: a first split starts from an empty mask, a ticket's from the saved one
            if (optOn && lane == 0) opt->done[wave] = myTicket ? *reinterpret_cast<const u64 *>(stBase + (long long)mySid * p.stateStride + offDone) : 0ull;
            if (!lazyNode) {
                // the hypothesis was kept when it was found: its state is in registers by now
                if (lane < D) {
                    nd.u[lane] = ldU;
                    nd.v[lane] = ldV;
                    nd.r4c[lane] = (unsigned char)ldR;
                    nd.c4r[lane] = (unsigned char)ldC;
                }
                if (lane == 0) {
                    nd.forb[0] = ldForb;
                    nd.gain[0] = ldGain;
                    nd.info[0] = ldA;
                    nd.info[1] = mySid;
                }
            } else {
                // re-solve candidate mySel in full from its parent's saved state
                const u32 meta = (u32)uni32((int)PM[mySel]) & META_MASK;
                const int par = (int)(meta >> 8), col = (int)(meta & 255u);
                const unsigned char *st = stBase + (long long)par * p.stateStride;
                const double *sd = reinterpret_cast<const double *>(st);
                double v = 0.0;
                int r4c = -1, c4r = -1;
                if (lane < D) {
                    nd.u[lane] = sd[lane];
                    v = sd[DS + lane];
                    r4c = st[offR4C + lane];
                    c4r = st[offC4R + lane];
                }
                const u64 forbP = uni64(*reinterpret_cast<const u64 *>(st + offTail));
                const int aP = uni32(*reinterpret_cast<const int *>(st + offTail + 16));
                const int fr = __builtin_amdgcn_readlane(r4c, col);
                const u64 cand = __ballot(lane < D && c4r >= col);
                const u64 forbm = (col == aP) ? forbP : bit64(fr);
                c4r = (lane == fr) ? -1 : c4r;
                r4c = (lane == col) ? -1 : r4c;
                double spc, delta;
                int pred, sink = 0;
                u64 scanned;
                const int rc = dijkstra<false>(Cs, LDC, nd.u, rl, lane, v, c4r, cand, forbm, col, INF, spc, pred, scanned,
                                               delta, sink, 0.0, 0, parkFrom);
                if (rc == 0) dual_update_flip(nd.u, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, col);
                const double g = serial_gain(Cs, LDC, lane, r4c, M, gainW, colOf);
                const u64 forbN = forbm | bit64(__builtin_amdgcn_readlane(r4c, col));  // cpp:362
                save_node(nd, mySid, v, r4c, c4r, forbN, g, col);
                if (lane == 0 && rc != 0) ctrl->stop = 2;  // cannot happen: the candidate was solved before
            }
        }
        KB_T(tA1);
        KB_ACC(3, tA1 - tA0);     // [3] select / emit / re-solve busy
        __syncthreads();
        KB_ACC(12, __builtin_readcyclecounter() - tA1);  // [12] wait at the barrier after A
    }
    const int stopCode = uni32(ctrl->stop);
    if (RELAY && stopCode == 4) {  // relay: this piece's share of the solutions is out (phase D saw it): the next workgroup goes on from here
        // hand over: the whole LDS (pool, nodes, tile, control, minima: everything a round starts from), the round number in it
        if (tid == 0) ctrl->relayRound = roundNo;
        __syncthreads();
#ifdef KB_PROFILE
        if (p.prof && threadIdx.x == 0) p.prof[(long long)gridDim.x * gridDim.y * 16 + ((long long)piece * gridDim.x + blockIdx.x) * 5 + 4] = wall_clock64();  // the rounds are over
#endif
        uint4 *dst = reinterpret_cast<uint4 *>(p.relayBuf + (long long)blk * p.relayStride);
        const uint4 *src = reinterpret_cast<const uint4 *>(smem);
        for (int i = tid; i < L.total / 16; i += NT) dst[i] = src[i];
        // every wave's stores -- the image's and, from the rounds, the hypothesis states' -- have left the CU before the barrier
        // (a workgroup barrier alone does not wait for them), then one lane writes the XCD's L2 back and, once THAT is done,
        // raises the flag (the wait is spelled in asm: the compiler drops its own behind a release fence when it believes the
        // wave has nothing outstanding)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.relayFlag + blk, (unsigned)piece + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        relay_depart();
#ifdef KB_PROFILE
        profAcc[13] = __builtin_readcyclecounter() - profT0;     // [13] whole kernel (this wave)
        if (p.prof && threadIdx.x == 0)
            p.prof[(long long)gridDim.x * gridDim.y * 16 + ((long long)piece * gridDim.x + blockIdx.x) * 5 + 1] = wall_clock64();
        if (p.prof && lane == 0)
            for (int i = 0; i < 16; i++) atomicAdd(p.prof + (long long)blk * 16 + i, profAcc[i]);
#endif
        return;
    }
    const int nfAll = (stopCode == 2) ? -3 : uni32(ctrl->emitted);
    const int nf = nfAll > kTab ? kTab : nfAll;
    // ---- phase 3: outputs.  Slot s holds hypothesis slotSid[s]: widen its saved row4col / col4row (the slots that were not
    //      written during the rounds: those emitted in the last one) --------
    const int outDoneEnd = uni32(ctrl->outDone);
    for (int idx = tid + outDoneEnd * (N + M); idx < nf * (N + M); idx += NT) {
        const int s = idx / (N + M), j = idx - s * (N + M);
        const unsigned char *st = stBase + (long long)slotSid[s] * p.stateStride;
        if (j < M) put_index(p.row4col, (outBase + s) * p.ldCol + colOf[j], st[offR4C + j], tabI8);
        else if (p.col4row) {
            const int cv = st[offC4R + (j - M)];
            put_index(p.col4row, (outBase + s) * p.ldRow + (j - M), (rect && cv == 255) ? -1 : (cv < M ? (int)colOf[cv] : cv), tabI8);  // unassigned row (cpp:134)
        }
    }
    if (tid == 0) {
        p.nf[blk] = nf;
        if (p.pushed) p.pushed[blk] = ctrl->pushed;
    }
    // (relay: this matrix is finished -- the workgroups of its later pieces have nothing to do)
    if (RELAY && tid == 0) __hip_atomic_store(p.relayFlag + blk, 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    relay_depart();
#undef kTab
#ifdef KB_PROFILE
    profAcc[13] = __builtin_readcyclecounter() - profT0;  // [13] whole kernel (this wave)
    if (p.prof && threadIdx.x == 0)
        p.prof[(long long)gridDim.x * gridDim.y * 16 + ((long long)piece * gridDim.x + blockIdx.x) * 5 + 1] = wall_clock64();
    if (p.prof && lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(p.prof + (long long)blk * 16 + i, profAcc[i]);
#endif
}

// ------------------------------------------------- conditionCosts prologue
// conditionCosts (assignment.cpp:439-525): column minima (:450-458); a row is kept iff some entry is within
// 42 of its column's minimum (:462-474); kept rows are compacted in order, entries become cost - colMin or
// +inf beyond the gate (:476-496).  One wave per cost matrix walking the rows 64 at a time, so the RAW matrix
// may have any number of rows (every landmark of the map): only the conditioned matrix has to fit the solver.
__global__ void __launch_bounds__(64) condition_kernel(CondParams p)
{
    __shared__ double colMin[WIDE_MAX_DIM];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nR = p.nRow[b], nC = p.nCol[b];
    const double *C = p.cost + p.costOff[b];
    double *out = p.out + p.costOff[b];
    int *ridx = p.rowIdx + (long long)b * p.maxRow;
    const double INF = d_inf(), GATE = 42.0;  // assignment.cpp:9
    for (int c = 0; c < nC; c++) {
        double m = INF;
        for (int r = lane; r < nR; r += 64) m = min_keep(m, C[(long long)c * nR + r]);
        m = wave_min_f64(m);
        if (lane == 0) colMin[c] = m;
    }
    __syncthreads();
    int g = 0;  // kept rows
    for (int r0 = 0; r0 < nR; r0 += 64) {
        const int r = r0 + lane;
        bool good = false;
        for (int c = 0; c < nC; c++)
            if (r < nR && C[(long long)c * nR + r] <= colMin[c] + GATE) good = true;
        g += __popcll(__ballot(good));
    }
    int base = 0;
    for (int r0 = 0; r0 < nR; r0 += 64) {
        const int r = r0 + lane;
        bool good = false;
        for (int c = 0; c < nC; c++)
            if (r < nR && C[(long long)c * nR + r] <= colMin[c] + GATE) good = true;
        const u64 mask = __ballot(good);
        const int nr = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (good) {
            ridx[nr] = r;
            for (int c = 0; c < nC; c++) {
                const double x = C[(long long)c * nR + r];
                out[(long long)c * g + nr] = (x <= colMin[c] + GATE) ? (x - colMin[c]) : INF;
            }
        }
        base += __popcll(mask);
    }
    if (lane == 0) {
        p.goodRows[b] = g;
        if (p.condL) p.condL[b] = g - nC;  // assignment.cpp:60
    }
}

// ------------------------------------------------- association weights epilogue
// assignmentProb accumulate / normalise (assignment.cpp:616-648) and its
// single-column fast path (assignment.cpp:554-570).  One wave per problem,
// lane = measurement (column); solutions are accumulated in ascending order
// exactly as the reference loop does, so only exp() itself can differ.
// With rowIdx != nullptr the problem was conditioned first and the result is
// scattered back to the original landmark numbering (getAssignmentProbs,
// assignment.cpp:68-74): output row stride nLout[b] + 1.
__global__ void __launch_bounds__(256) weights_kernel(WeightParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    constexpr int NT = 256;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nL = p.nL[b], nM = p.nM[b];
    const int nLo = p.rowIdx ? p.nLout[b] : nL;  // landmarks in the output numbering
    const int *ridx = p.rowIdx ? p.rowIdx + (long long)b * p.maxRow : nullptr;
    double *probs = p.probs + p.probOff[b];
    const double GATE = 42.0;  // assignment.cpp:9
    for (int i = tid; i < nM * (nLo + 1); i += NT) probs[i] = 0.0;
    __syncthreads();
    if (nL < 0) return;  // fewer kept rows than measurements: undefined in the reference (size_t underflow, :60)
    if (nM > 1 && p.nf[b] < 0) return;  // the conditioned matrix did not fit the solver: probabilities stay 0, nf < 0
    if (nM == 1) {
        const double *cost = p.cost + p.costOff[b];
        if (tid == 0) {
            double norm = 0.0;
            for (int i = 0; i <= nL; i++)
                if (cost[i] < GATE) norm += exp(-cost[i]);
            norm = 1.0 / norm;
            for (int i = 0; i <= nL; i++) {
                const double q = (cost[i] < GATE) ? exp(-cost[i]) : 0.0;
                probs[(i >= nL) ? nLo : (ridx ? ridx[i] : i)] = q * norm;
            }
        }
        return;
    }
    const int nf = (p.kUse > 0 && p.nf[b] > p.kUse) ? p.kUse : p.nf[b];  // (kUse: the tables hold more solutions than are weighed -- a completed tie level)
    const double *gain = p.gain + (long long)b * p.k;
    const int *r4c = p.row4col + (long long)b * p.k * p.maxCol;
    const double best = gain[0];
    // The sums run in the reference's order -- solutions ascending, `total` and every probs[col][row] sequentially -- but
    // everything around them is parallel: per chunk of solutions the weights exp(best - g) are computed by all threads,
    // the chunk's row maps come into LDS in one coalesced pass, then thread = column walks the chunk (LDS reads only),
    // accumulating in LDS when the [nM][nL+1] table fits (ldsAcc), else in the output block.
    double *wts = reinterpret_cast<double *>(wsm);                       // [chunk]
    int *rows = reinterpret_cast<int *>(wsm + p.chunk * 8);               // [chunk][nM]
    double *acc = reinterpret_cast<double *>(wsm + p.chunk * 8 + (size_t)p.chunk * p.maxCol * 4);  // [nM][nL+1] (ldsAcc)
    const bool ldsAcc = p.ldsAcc != 0 && (size_t)nM * (nL + 1) * 8 <= (size_t)p.accBytes;
    if (ldsAcc)
        for (int i = tid; i < nM * (nL + 1); i += NT) acc[i] = 0.0;
    double total = 0.0;
    for (int s0 = 0; s0 < nf; s0 += p.chunk) {
        const int ns = (nf - s0) < p.chunk ? (nf - s0) : p.chunk;
        __syncthreads();
        for (int s = tid; s < ns; s += NT) {
            const double g = gain[s0 + s];
            wts[s] = (p.gate && !(best + GATE > g)) ? -1.0 : exp(best - g);  // :622-626 (bruteForceProb sums every solution)
        }
        for (int i = tid; i < ns * nM; i += NT) {
            const int s = i / nM, m = i - s * nM;
            rows[i] = r4c[(long long)(s0 + s) * p.maxCol + m];
        }
        __syncthreads();
        if (tid == 0)
            for (int s = 0; s < ns; s++)
                if (wts[s] >= 0.0) total += wts[s];
        for (int m = tid; m < nM; m += NT) {
            for (int s = 0; s < ns; s++) {
                const double w = wts[s];
                if (w < 0.0) continue;
                const int r = rows[s * nM + m];
                if (ldsAcc) acc[m * (nL + 1) + ((r >= nL) ? nL : r)] += w;                         // :633-638
                else probs[m * (nLo + 1) + ((r >= nL) ? nLo : (ridx ? ridx[r] : r))] += w;
            }
        }
    }
    __syncthreads();
    double *tot = wts;
    if (tid == 0) tot[0] = total;
    __syncthreads();
    const double norm = 1.0 / tot[0];  // :643
    if (ldsAcc) {
        for (int i = tid; i < nM * (nL + 1); i += NT) {
            const int m = i / (nL + 1), r = i - m * (nL + 1);
            probs[m * (nLo + 1) + ((r >= nL) ? nLo : (ridx ? ridx[r] : r))] = acc[i] * norm;  // scatter back (:68-74)
        }
    } else {
        for (int i = tid; i < nM * (nLo + 1); i += NT) probs[i] *= norm;
    }
}

// ------------------------------------------------------------------- launchers
template <int NW, int EPT, bool RELAY>
static hipError_t launch_nw_ept(const Params &p, int B, hipStream_t stream)
{
    const Lds L = lds_layout(p.maxRow, p.k, p.spec, NW);
    static const int pad = getenv("KBEST_LDS_PAD") ? atoi(getenv("KBEST_LDS_PAD")) : 0;  // residency experiments only
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_kernel<NW, EPT, RELAY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, L.total + pad);
    if (e != hipSuccess) return e;
    // (relay launches: gridDim.y workgroups per matrix; the order of their arrival is the order of the pieces)
    hipLaunchKernelGGL((kbest_kernel<NW, EPT, RELAY>), dim3(B, RELAY ? p.relayP : 1), dim3(NW * 64), L.total + pad, stream, p);
    return hipGetLastError();
}

template <int NW>
static hipError_t launch_nw(const Params &p, int B, hipStream_t stream)
{
    // (relay launches: the shapes relay_shape_ok() names, one instantiation each)
    if (p.relayP > 1) {
        if constexpr (NW == 4 || NW == 8 || NW == 12)
            return (p.k <= NW * 64) ? launch_nw_ept<NW, 1, true>(p, B, stream) : launch_nw_ept<NW, 4, true>(p, B, stream);
        return hipErrorInvalidValue;
    }
    return (p.k <= NW * 64) ? launch_nw_ept<NW, 1, false>(p, B, stream) : launch_nw_ept<NW, 4, false>(p, B, stream);
}

hipError_t launch_kbest(const Params &p, int B, int nWaves, hipStream_t stream)
{
    switch (nWaves) {
    case 1: return launch_nw<1>(p, B, stream);
    case 2: return launch_nw<2>(p, B, stream);
    case 4: return launch_nw<4>(p, B, stream);
    case 12: return launch_nw<12>(p, B, stream);
    case 16: return launch_nw<16>(p, B, stream);
    default: return launch_nw<8>(p, B, stream);
    }
}

hipError_t launch_condition(const CondParams &p, int B, hipStream_t stream)
{
    hipLaunchKernelGGL(condition_kernel, dim3(B), dim3(64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_weights(const WeightParams &p0, int B, hipStream_t stream)
{
    WeightParams p = p0;
    // LDS: weights and row maps of one chunk of solutions, and -- when it fits -- the accumulator table
    int chunk = 8192 / (p.maxCol > 0 ? p.maxCol : 1);
    chunk = chunk > 256 ? 256 : (chunk < 8 ? 8 : chunk);
    const size_t base = (size_t)chunk * 8 + (size_t)chunk * p.maxCol * 4;
    const size_t accWant = (size_t)p.maxCol * (size_t)(p.solveRows > 0 ? p.solveRows : 1) * 8;
    p.chunk = chunk;
    p.ldsAcc = (base + accWant <= 96 * 1024) ? 1 : 0;
    p.accBytes = p.ldsAcc ? (long long)accWant : 0;
    const size_t lds = ((base + (p.ldsAcc ? accWant : 0)) + 15) & ~(size_t)15;
    static std::atomic<size_t> granted{0};
    if (lds > granted.load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(weights_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted.store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(weights_kernel, dim3(B), dim3(256), lds, stream, p);
    return hipGetLastError();
}

}  // namespace kb
