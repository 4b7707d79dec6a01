// kbest_wide.hip -- the general-size form of the k-best enumeration for gfx950.
//
// kbest_engine.hip keeps a whole problem in LDS and covers numRow <= 64 with a
// candidate pool of a few thousand entries.  The reference has no such limits
// (kBest2D, shortestPathCPP.cpp:571-644, takes any numRow >= numCol and any k;
// bruteForceProb, assignment.cpp:868, asks for up to 20 000 assignments), so
// this kernel handles everything beyond them -- numRow up to 64 * R (R rows per
// lane, R <= 16: 1 024 rows) and any k -- with the SAME arithmetic in the same order:
//   * cost copy, duals, hypotheses and the candidate pool live in HBM work
//     space (L2-resident for the sizes in question), only the hypothesis a
//     wave is working on sits in LDS / registers;
//   * one workgroup per problem, persistent over a grid-stride loop of
//     problems, so the work space is bounded by the grid, not by the batch;
//   * per sweep the popped hypothesis' children (split, cpp:455-532) are solved
//     by the waves in parallel (shortestPathUpdateCPP, cpp:240-365), with the
//     same early termination against the pool's (k - emitted)-th gain as the
//     LDS kernel, then rank-merged into the sorted pool (ties in gain: pool
//     entries before new children, children by column);
//   * the order of operations is the reference's: pop the minimum, split it,
//     emit the new minimum (kBest2D cpp:607-634, kBest2DCutoff cpp:690-722).
// Gains are the serial column-order sum (calcGain cpp:59-80), the reduced cost
// is ((delta + C) - u) - v left to right (cpp:183, 313), the arg-min takes the
// lowest row index (cpp:191, 320): results are identical to the reference's.
//
// fp64 add/sub/compare only; compiled WITHOUT fast-math.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kbest_engine.h"
#include "kbest_wave.h"

namespace kb {

namespace {

struct WideCtrl {
    double cdelta;      // CDelta * numCol (cpp:583)
    double cutoffGain;  // workMem.cutoffGain (cpp:681/684)
    double cmax;        // largest finite shifted cost
    double gain0u;      // gainBest[0]
    int n;           // pool entries (entry 0 = the hypothesis emitted last, popped next)
    int cur;         // which of the two pool buffers is current
    int emitted;     // output slots filled
    int stop;        // 1: finished  2: internal error  3: infeasible root
    int pushed;
    int nFree;       // free state slots (stack in freeList)
    int nChild;      // children that survived this sweep
    int nextTicket;  // work queue over the children of the round
    double t0;       // a-priori threshold (wide_apriori_threshold), +inf when unknown
    int nsel;        // hypotheses split in this round
    int nEmit;       // pool entries emitted in this round (the first nEmit)
    int lastSel;     // pool index of the last selected entry: every not yet split entry up to it is selected
    int nextB;       // the problem this workgroup takes next (the batch's queue)
    int selIdx[WIDE_MAX_SPEC], selSid[WIDE_MAX_SPEC], selA[WIDE_MAX_SPEC], selOff[WIDE_MAX_SPEC + 1];  // the selected
    double selG[WIDE_MAX_SPEC];  // entries: pool index, state slot, active column, first ticket, gain
};
static_assert(sizeof(WideCtrl) <= WIDE_CTRL_BYTES, "WideCtrl must fit the LDS slot reserved by wide_lds_layout");
constexpr int WIDE_SPLIT = 0x40000000;      // pool entry flag: children already generated
constexpr int WIDE_SID_MASK = 0x3FFFFFFF;

// Shortest augmenting path from column `start`; lane owns rows lane + 64*i.  Same contract as dijkstra<> of
// kbest_engine.hip (cpp:168-226 / cpp:297-356) with per-lane bit sets: cand bit i = row lane+64i still in
// Row2Scan, forb = rows skipped while the start column itself is scanned (cpp:310).
// Returns 0 = path found, 1 = infeasible (cpp:197, 327), 2 = abandoned (delta beyond `bound`, EARLY only).
template <int R, bool EARLY>
__device__ __forceinline__ int wide_dijkstra(const double *Cw, int D, const double *uW, const int *c4rW, int lane,
                                             const double (&v)[R], u32 cand, u32 forb, int start, double bound,
                                             double (&spc)[R], int (&pred)[R], u32 &scannedOut, double &deltaOut,
                                             int &sinkOut, int M = 0x7fffffff, int *stepsOut = nullptr, int freed = -1)
{
    // `freed`: the row the child took from its start column (cpp:277-278).  uW / c4rW may then be the PARENT's arrays,
    // shared and read-only: the only entry of col4row that differs in the child is that row's (-1: it is the free row).
    // M < D (children of a rectangular problem): rows on the zero-padded columns M .. D-1 ("parked") all carry the same
    // dual, and so do those columns, in every dual-feasible solution -- once the search has settled ONE parked row at
    // distance d every other parked row is at distance d too and scanning their columns changes nothing (kbest_small.hip,
    // file header).  So when the first parked row is settled, all of them are: one step instead of one per parked row.
    const double INF = d_inf();
    int rowOff[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        spc[i] = INF;
        pred[i] = 0;
        const int r = lane + 64 * i;
        rowOff[i] = r < D ? r : D - 1;  // rows beyond D are never candidates: any address inside the column will do
    }
    u32 act = cand & ~forb, scanned = 0;
    int cur = uni32(start), parkThr = uni32(M), steps = 0;
    bound = __hiloint2double(uni32(__double2hiint(bound)), uni32(__double2loint(bound)));
    double delta = 0.0;
    for (;;) {
        // the plain steps: straight-line relaxation (selects, no divergent blocks), one exit test
        int cc, closest;
        do {
            steps++;
            const double ucur = uW[cur];
            const double *col = Cw + (long long)cur * D;
            double best = INF;
            int brow = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < R; i++) {
                const bool on = ((act >> i) & 1u) != 0;
                const double rc = ((delta + col[rowOff[i]]) - ucur) - v[i];  // cpp:183 / cpp:313, left to right
                const bool better = on & (rc < spc[i]);                      // strict '<': cpp:185, 314
                spc[i] = better ? rc : spc[i];
                pred[i] = better ? cur : pred[i];
                const bool low = on & (spc[i] < best);                       // lowest row first: cpp:191, 320
                best = low ? spc[i] : best;
                brow = low ? lane + 64 * i : brow;
            }
            // wave arg-min.  Reduced costs are non-negative up to rounding, and for non-negative doubles the high word is
            // an order-preserving key: one integer DPP chain finds it, and when a single lane holds it (four steps in
            // five) that lane's row is the answer.  Otherwise -- a negative candidate, or several lanes on the same high
            // word (mostly exact zeros on tight arcs) -- the full fp64 minimum and the lowest row among its holders.
            const int bhi = __double2hiint(best);
            const int mhi = wave_min_i32(bhi);
            const u64 eq = __ballot(bhi == mhi);
            double m;
            if (mhi >= 0 && (eq & (eq - 1)) == 0) {
                const int ln = __builtin_ctzll(eq);
                m = __hiloint2double(mhi, __builtin_amdgcn_readlane(__double2loint(best), ln));
                closest = __builtin_amdgcn_readlane(brow, ln);
            } else {
                m = wave_min_f64(best);
                closest = wave_min_i32(best == m ? brow : 0x7fffffff);
            }
            if (!(m < INF) || (EARLY && m > bound)) {  // infeasible (cpp:197, 327) / beyond the bound
                scannedOut = scanned;
                if (stepsOut) *stepsOut = steps;
                return (m < INF) ? 2 : 1;
            }
            delta = m;
            const u32 bit = (lane == (closest & 63)) ? (1u << (closest >> 6)) : 0u;
            cand &= ~bit;
            scanned |= bit;
            act = cand;
            cc = (closest == freed) ? -1 : uni32(c4rW[closest]);
            cur = cc;
        } while (cc >= 0 && cc < parkThr);
        if (cc < 0) { sinkOut = closest; break; }
        // the first parked row is settled: so are all the others (at the same distance, through column cc)
        parkThr = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < R; i++) {
            if (((cand >> i) & 1u) && c4rW[rowOff[i]] >= M) {
                spc[i] = delta;  // (what the relaxation through column cc gives them: equal duals, zero costs)
                pred[i] = cc;
                cand &= ~(1u << i);
                scanned |= 1u << i;
            }
        }
        act = cand;
    }
    if (stepsOut) *stepsOut = steps;
    scannedOut = scanned;
    deltaOut = delta;
    return 0;
}

// updateDualAndAugment (cpp:82-117) on the wave's working hypothesis (u, col4row, row4col in LDS; v in registers)
template <int R>
__device__ __forceinline__ void wide_update(double *uW, int *c4rW, int *r4cW, int *predW, int lane, double (&v)[R],
                                            const double (&spc)[R], const int (&pred)[R], u32 scanned, double delta,
                                            int sink, int start, int D)
{
#pragma unroll
    for (int i = 0; i < R; i++) {
        if ((scanned >> i) & 1u) {
            const int r = lane + 64 * i;
            predW[r] = pred[i];
            if (r != sink) {  // scanned columns other than start: cpp:96-99
                const int c = c4rW[r];
                uW[c] = uW[c] + delta - spc[i];
            }
            v[i] = v[i] - delta + spc[i];  // cpp:102-106
        }
    }
    if (lane == 0) uW[start] = uW[start] + delta;  // cpp:92
    wave_fence();
    if (lane == 0) {  // cpp:108-116
        int r = sink, c, guard = 0;
        do {
            c = predW[r];
            c4rW[r] = c;
            const int nxt = r4cW[c];
            r4cW[c] = r;
            r = nxt;
        } while (c != start && ++guard < D);
    }
    wave_fence();
}

// calcGain (cpp:59-80): serial left-to-right sum over the M real columns, from 0.0
template <int R>
// `posOf` (the enumeration runs in a column order of its own, phase 1b): the maps and the cost copy are indexed by POSITION,
// posOf[c] is the position of the reference's column c -- the terms are fetched per reference column, so the chain adds them
// in the reference's order.
__device__ __forceinline__ double wide_gain(const double *Cw, int D, int M, const int *r4cW, int lane,
                                            const unsigned short *posOf = nullptr)
{
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        if (64 * i < M) {
            const int c = lane + 64 * i;
            double t = 0.0;  // columns >= M add +0.0: exact for the non-negative partial sums
            if (c < M) {
                const int pc = posOf ? (int)posOf[c] : c;
                t = Cw[r4cW[pc] + (long long)pc * D];
            }
            const int tlo = __double2loint(t), thi = __double2hiint(t);
            for (int j0 = 0; j0 < 64 && 64 * i + j0 < M; j0 += 8) {
                double term[8];
#pragma unroll
                for (int e = 0; e < 8; e++)
                    term[e] = __hiloint2double(__builtin_amdgcn_readlane(thi, j0 + e), __builtin_amdgcn_readlane(tlo, j0 + e));
#pragma unroll
                for (int e = 0; e < 8; e++) acc = acc + term[e];
            }
        }
    }
    return acc;
}

}  // namespace

// A-priori threshold on the k-th best gain (the idea and the argument: apriori_threshold, kbest_engine.hip): every child of
// the root differs from the optimum by one alternating path; children that move disjoint rows combine into further known
// assignments (singles, pairs of the 64 cheapest, triples of the 16, quadruples of the 8 cheapest); the (k-1)-th smallest
// of their costs bounds the k-th best gain while the pool has no threshold yet.  Row sets are W words here (row r = bit
// r % 64 of word r / 64, W = rows per lane).  atoms: per column c (1 + W) words -- cost bits, rows moved --, then the root's gain.
// sd / sm / cnt: LDS scratch (64 doubles, 64 x W words, 32 ints); cost: LDS, >= M doubles.
template <int W, int NT>
__device__ __attribute__((noinline)) void wide_apriori_threshold(double *sd, u64 *sm, int *cnt, double *cost, const u64 *atoms,
                                                                  int M, int k, double *t0Out)
{
    constexpr int SLOTS = (4096 + NT - 1) / NT;
    const double INF = d_inf();
    const int tid = threadIdx.x, lane = tid & 63;
    for (int c = tid; c < M; c += NT) cost[c] = __longlong_as_double((long long)atoms[(long long)c * (1 + W)]);
    if (tid < 64) sd[tid] = INF;
    if (tid < 32) cnt[tid] = 0;
    __syncthreads();
    for (int c = tid; c < M; c += NT) {  // the 64 cheapest atoms, sorted (ties by column)
        const double dl = cost[c];
        int rank = 0;
        for (int j = 0; j < M; j++) {
            const double dj = cost[j];
            rank += (dj < dl || (dj == dl && j < c)) ? 1 : 0;
        }
        if (rank < 64 && dl < INF) {
            sd[rank] = dl;
#pragma unroll
            for (int w = 0; w < W; w++) sm[rank * W + w] = atoms[(long long)c * (1 + W) + 1 + w];
        }
    }
    __syncthreads();
    const int nA = __popcll(__ballot(sd[lane] < INF));
    if (nA < 2) return;  // (uniform)
    auto disjoint = [&](int i, int j) {
        u64 x = 0;
#pragma unroll
        for (int w = 0; w < W; w++) x |= sm[i * W + w] & sm[j * W + w];
        return x == 0ull;
    };
    double val[3 * SLOTS + 1];
    val[3 * SLOTS] = tid < 64 ? sd[tid] : INF;
#pragma unroll
    for (int e = 0; e < SLOTS; e++) {
        const int idx = tid + e * NT;
        {
            const int i = idx >> 6, j = idx & 63;
            double x = INF;
            if (idx < 4096 && i < j && j < nA && disjoint(i, j)) x = sd[i] + sd[j];
            val[e] = x;
        }
        {
            const int i = idx >> 8, j = (idx >> 4) & 15, l = idx & 15;
            double x = INF;
            if (idx < 4096 && i < j && j < l && l < nA && disjoint(i, j) && disjoint(i, l) && disjoint(j, l)) x = (sd[i] + sd[j]) + sd[l];
            val[SLOTS + e] = x;
        }
        {
            const int i = idx >> 9, j = (idx >> 6) & 7, l = (idx >> 3) & 7, q = idx & 7;
            double x = INF;
            if (idx < 4096 && i < j && j < l && l < q && q < nA && disjoint(i, j) && disjoint(i, l) && disjoint(i, q) &&
                disjoint(j, l) && disjoint(j, q) && disjoint(l, q))
                x = ((sd[i] + sd[j]) + sd[l]) + sd[q];
            val[2 * SLOTS + e] = x;
        }
    }
    auto total_le = [&](double x, int step) -> int {
        int n = 0;
#pragma unroll
        for (int e = 0; e <= 3 * SLOTS; e++) n += (val[e] <= x) ? 1 : 0;
        int w = 0;
#pragma unroll
        for (int bit = 0; bit < 5; bit++) w += __popcll(__ballot((n >> bit) & 1)) << bit;
        if (lane == 0 && w) atomicAdd(&cnt[step], w);
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(cnt[step]);
    };
    const double top = 2.0 * sd[nA - 1];
    double hi = 2.0 * sd[nA > 16 ? 15 : nA - 1], lo = 0.0;
    int step = 0;
    for (;;) {
        if (hi > top) hi = top;
        if (total_le(hi, step++) >= k - 1) break;
        if (hi >= top || step >= 12) return;  // fewer than k - 1 known assignments: no threshold
        lo = hi;
        hi = 2.0 * hi;
    }
    for (int it = 0; it < 8; it++) {
        const double mid = 0.5 * (lo + hi);
        if (total_le(mid, step++) >= k - 1) hi = mid; else lo = mid;
    }
    if (tid == 0) *t0Out = __longlong_as_double((long long)atoms[(long long)M * (1 + W)]) + hi;
}


#ifdef KB_PROFILE
#define KW_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define KW_ACC(slot, expr) do { profAcc[slot] += (unsigned long long)(expr); } while (0)
#else
#define KW_T(var) do { } while (0)
#define KW_ACC(slot, expr) do { } while (0)
#endif

// up to 128 rows the working set fits 80 VGPRs: three 8-wave workgroups per CU instead of two
template <int R, bool TILE, int NWV>
__global__ void __launch_bounds__(NWV * 64, (R >= 16 ? 1 : (NWV == 16 ? 4 : (R <= 2 ? 6 : 4)))) kbest_wide_kernel(WideParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NWV * 64;
    const double INF = d_inf();
    int tid = threadIdx.x;   // (not const: made opaque once per round, see the round loop)
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int DS = p.maxRow;
    const WideLds L = wide_lds_layout(DS, p.maxCol, TILE, NWV, p.spec);
    double *uW = reinterpret_cast<double *>(smem + L.offWave + (size_t)wave * L.waveStride);
    int *c4rW = reinterpret_cast<int *>(uW + DS);
    int *r4cW = c4rW + DS;
    int *predW = r4cW + DS;
    double *childG = reinterpret_cast<double *>(smem + L.offChildG);
    int *childS = reinterpret_cast<int *>(smem + L.offChildS);
    int *childC = reinterpret_cast<int *>(smem + L.offChildC);
    double *samp = reinterpret_cast<double *>(smem + L.offSample);  // every 64th gain of the current pool (merge)
    unsigned char *nodeBase = smem + L.offNode;  // the hypotheses being split in this round: copies of their saved states
    double *red = reinterpret_cast<double *>(smem + L.offRed);
    WideCtrl *ctrl = reinterpret_cast<WideCtrl *>(smem + L.offCtrl);

    // work space of this workgroup (whole 128-byte lines, never shared with another workgroup)
    const long long ws = blockIdx.x;
    double *Cw;  // shifted, zero-padded square cost copy: LDS when it fits (TILE), else this workgroup's HBM slot
    if constexpr (TILE) Cw = reinterpret_cast<double *>(smem + L.offTile);
    else Cw = p.Cw + ws * p.cwStride;
    unsigned char *stBase = p.states + ws * (long long)p.statesPerProblem * p.stateStride;
    double *poolG = p.poolG + ws * 2 * p.poolStride;
    int *poolS = p.poolS + ws * 2 * p.poolStride;
    int *freeList = p.freeList + ws * p.freeStride;
    const int atomSlots = wide_atom_slots(p.maxRow, p.maxCol);
    const int S = p.statesPerProblem - atomSlots;  // the last slots hold the atoms of the a-priori threshold
    const int k = p.k;
    const bool maximize = p.maximize != 0, useCut = p.useCutoff != 0;
    const bool prune = (p.flags & KBEST_FLAG_NO_PRUNE) == 0;
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    // saved hypothesis: u[DS] v[DS] (fp64) | row4col[DS] col4row[DS] (i32) | forbidden rows (u32 per lane) | gain, activeCol
    const long long offV = 8LL * DS, offR4C = 16LL * DS, offC4R = 20LL * DS, offForb = 24LL * DS, offTail = 24LL * DS + 256;

    // The grid is as many workgroups as the chip holds; a batch larger than that is a QUEUE: a workgroup that has finished a problem
    // takes the next one nobody has taken (one atomic; the first gridDim.x problems need none), so that the launch ends within one
    // problem's time of the moment the queue runs dry -- with a fixed stride it ended with the workgroup whose problems add up to
    // the most.  The last workgroup to leave puts the queue's two words back to zero.
    auto next_problem = [&](int bNow) -> int {
        if (!p.queue) return bNow + (int)gridDim.x;
        __syncthreads();
        if (tid == 0) ctrl->nextB = (int)gridDim.x + (int)atomicAdd(p.queue, 1u);
        __syncthreads();
        return uni32(ctrl->nextB);
    };
    for (int b = blockIdx.x; b < p.B; b = next_problem(b)) {
        __syncthreads();  // the previous problem of this workgroup is finished
        const int N = p.nRow ? p.nRow[b] : p.maxRow;
        const int M = p.nCol ? p.nCol[b] : p.maxCol;
        if (N >= 1 && M >= 1 && N >= M && N < p.minRows) continue;  // the LDS kernel's share of a mixed batch
        // exact ties (kbest_ties.h): the tables hold p.kTab slots (k, or k - 1: the k-th solution is enumerated for its gain only)
        if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol || N > 64 * R) {  // undefined in the reference
            if (tid == 0) p.nf[b] = -1;
            continue;
        }
        const int D = N;
        const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.ldRow * p.ldCol);
        // a-priori threshold (top of round 1): needs the whole problem's enumeration (no root-subtree sharding), pruning, and
        // LDS scratch in the node area: 64 x (1 + R) words + 32 counters, and M doubles in the children's gain list
        u64 *atoms = reinterpret_cast<u64 *>(p.states + (ws * (long long)p.statesPerProblem + S) * p.stateStride);
        const bool t0On = prune && k >= 3 && p.rootColStride <= 1 && !(p.flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_T0)) &&
                          p.spec * L.nodeStride >= 64 * (1 + R) * 8 + 128 && p.spec * p.maxCol >= M;
        if (t0On)
            for (int c = tid; c < M; c += NT) atoms[(long long)c * (1 + R)] = 0x7ff0000000000000ull;  // +inf: no such child
        const long long outBase = (long long)b * p.kTab;
#ifdef KB_PROFILE
        unsigned long long profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long profT0 = __builtin_readcyclecounter();
#endif

        // ---- phase 0: makeCostMatrixSafe + zero padding (cpp:534-569, 582-585) ----
        double cdelW = 0.0;  // the shift of the cost copy (phase 1b writes its columns again in another order)
        unsigned short *colOf = reinterpret_cast<unsigned short *>(smem + L.offPerm), *posOf = colOf + p.maxRow;
        for (int i = tid; i < D; i += NT) { colOf[i] = (unsigned short)i; posOf[i] = (unsigned short)i; }
        {
            double mn = INF;
            for (int i = tid; i < N * M; i += NT) {
                double x = Cg[i];
                x = maximize ? -x : x;
                mn = min_keep(mn, x);
            }
            mn = wave_min_f64(mn);
            if (lane == 0) red[wave] = mn;
            __syncthreads();
            mn = red[0];
            for (int w = 1; w < NWV; w++) mn = min_keep(mn, red[w]);
            const double cdel = maximize ? -mn : mn;
            cdelW = cdel;
            __syncthreads();
            double cm = 0.0;
            for (int i = tid; i < D * D; i += NT) {
                double val = 0.0;
                if (i < N * M) {
                    const double x = Cg[i];
                    val = maximize ? (-x + cdel) : (x - cdel);  // cpp:558 / cpp:564
                    if (val != val) val = INF;                   // inf - inf: behaves like +inf in every comparison
                    if (val < INF && val > cm) cm = val;
                }
                Cw[i] = val;
            }
            cm = -wave_min_f64(-cm);
            if (lane == 0) red[wave] = cm;
            for (int i = tid; i < S - 1; i += NT) freeList[i] = S - 1 - i;  // slot 0 is the root's
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NWV; w++) cm = red[w] > cm ? red[w] : cm;
                ctrl->cmax = cm;
                ctrl->cdelta = cdel * (double)M;  // cpp:583
                ctrl->stop = 0;
                ctrl->pushed = 0;
                ctrl->n = 0;
                ctrl->cur = 0;
                ctrl->emitted = 0;
                ctrl->nFree = S - 1;
                ctrl->nChild = 0;
                ctrl->nextTicket = 0;
                ctrl->t0 = INF;
                if (p.tieGain) p.tieGain[b] = __longlong_as_double(0x7ff8000000000000LL);  // exact ties (kbest_ties.h): no solution behind the tables (yet)
            }
            __syncthreads();
        }

        KW_ACC(0, __builtin_readcyclecounter() - profT0);  // [0] cost copy
        KW_T(tRoot);
        auto store_state = [&](int sid, const double (&v)[R], u32 forb, double gain, int activeCol) {
            unsigned char *st = stBase + (long long)sid * p.stateStride;
            double *su = reinterpret_cast<double *>(st), *sv = reinterpret_cast<double *>(st + offV);
            int *sr = reinterpret_cast<int *>(st + offR4C), *sc = reinterpret_cast<int *>(st + offC4R);
#pragma unroll
            for (int i = 0; i < R; i++) {
                const int r = lane + 64 * i;
                if (r < D) { su[r] = uW[r]; sv[r] = v[i]; sr[r] = r4cW[r]; sc[r] = c4rW[r]; }
            }
            reinterpret_cast<u32 *>(st + offForb)[lane] = forb;
            if (lane == 0) {
                *reinterpret_cast<double *>(st + offTail) = gain;
                *reinterpret_cast<int *>(st + offTail + 8) = activeCol;
            }
        };

        // ---- phase 1: root LAP (shortestPathCPP, cpp:119-238) on wave 0 -> state 0, output slot 0 ----
        if (wave == 0) {
            double v[R], spc[R];
            int pred[R];
            u32 all = 0;
#pragma unroll
            for (int i = 0; i < R; i++) {
                const int r = lane + 64 * i;
                v[i] = 0.0;
                if (r < D) { uW[r] = 0.0; c4rW[r] = -1; r4cW[r] = -1; all |= 1u << i; }
            }
            wave_fence();
            bool bad = false;
            // The REAL columns by shortest augmenting paths (cpp:139-230); the zero-padded ones (cpp:582-585) are filled
            // directly afterwards: every row that is still free has v = 0 (only scanned rows change v, a scanned free row
            // is the sink: unchanged), every other row v <= 0, so "padded column M + j <- the j-th free row, u = 0" is tight
            // on the assigned arcs and leaves every reduced cost 0 - 0 - v[r] >= 0 -- an optimal solution of the padded
            // problem without the N - M augmentations over tied zero columns (about N^2 / 2 Dijkstra steps on a map of
            // 100 landmarks x 20 measurements).  Which padded column a free row sits on is immaterial (SURVEY 8(a)
            // quirk 6); the duals are another optimal pair than the reference's, which no output depends on.
            const int nAug = (p.flags & KBEST_FLAG_EXACT_ROOT) ? D : M;
            if (!(p.flags & KBEST_FLAG_EXACT_ROOT) && M == D) {
                // Square problems start from a column reduction and a row reduction (as the LDS kernels, kbest_engine.hip):
                // u[c] = min of column c, the row of that minimum goes to the lowest such column; then a row without a
                // column takes v[r] = min over c of (C[r,c] - u[c]) and that column if it is still free (the lowest row wins).
                // Reduced costs stay >= 0, assigned arcs are tight; the augmentations below run from the columns left over
                // -- a quarter of them -- with half the Dijkstra steps (128x128: 518 instead of 1 049).
                for (int c = 0; c < D; c++) {
                    const double *col = Cw + (long long)c * D;
                    double m = INF;
                    int am = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < R; i++) {
                        const int r = lane + 64 * i;
                        const double x = r < D ? col[r] : INF;
                        const bool better = x < m;  // (rows of one lane ascend: strict '<' keeps the lowest)
                        m = better ? x : m;
                        am = better ? r : am;
                    }
                    const double mm = wave_min_f64(m);
                    if (!(mm < INF)) continue;  // (uniform) a column without a finite entry: the search below reports it
                    const int row = wave_min_i32(m == mm ? am : 0x7fffffff);  // the lowest row among equal minima
                    if (lane == 0) {
                        uW[c] = mm;
                        if (c4rW[row] < 0) { c4rW[row] = c; r4cW[c] = row; }
                    }
                    wave_fence();
                }
                double m2[R];
                int a2[R];
#pragma unroll
                for (int i = 0; i < R; i++) { m2[i] = INF; a2[i] = 0; }
                for (int c = 0; c < D; c++) {
                    const double uc = uW[c];
                    const double *col = Cw + (long long)c * D;
#pragma unroll
                    for (int i = 0; i < R; i++) {
                        const int r = lane + 64 * i;
                        const double d = (r < D ? col[r] : INF) - uc;
                        const bool better = d < m2[i];  // strict '<': the lowest column among equal minima
                        m2[i] = better ? d : m2[i];
                        a2[i] = better ? c : a2[i];
                    }
                }
                int *owner = predW;  // (free until the first augmentation)
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    if (r < D) owner[r] = 0x7fffffff;
                }
                wave_fence();
                bool want[R];
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    want[i] = r < D && c4rW[r] < 0 && m2[i] < INF;
                    if (want[i]) {
                        v[i] = m2[i];
                        if (r4cW[a2[i]] < 0) atomicMin(&owner[a2[i]], r);
                        else want[i] = false;
                    }
                }
                wave_fence();
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    if (want[i] && owner[a2[i]] == r) { c4rW[r] = a2[i]; r4cW[a2[i]] = r; }
                }
                wave_fence();
            }
            for (int c = 0; c < nAug; c++) {
                if (uni32(r4cW[c]) >= 0) continue;  // assigned by the reductions above
                u32 scanned;
                double delta;
                int sink = 0;
                if (wide_dijkstra<R, false>(Cw, D, uW, c4rW, lane, v, all, 0u, c, INF, spc, pred, scanned, delta, sink)) {
                    bad = true;
                    break;
                }
                wide_update<R>(uW, c4rW, r4cW, predW, lane, v, spc, pred, scanned, delta, sink, c, D);
            }
            if (!bad && nAug < D) {
                int before = 0;
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    const bool fre = r < D && c4rW[r] < 0;
                    const u64 m = __ballot(fre);
                    if (fre) {
                        const int j = before + __popcll(m & ((1ull << lane) - 1ull));
                        c4rW[r] = M + j;
                        r4cW[M + j] = r;
                    }
                    before += __popcll(m);
                }
                wave_fence();
            }
            if (bad) {
                if (lane == 0) ctrl->stop = 3;
            } else {
                const double g = wide_gain<R>(Cw, D, M, r4cW, lane);  // (the root: before phase 1b, the reference's order)
                const int r0 = uni32(r4cW[0]);
                const u32 forb = (lane == (r0 & 63)) ? (1u << (r0 >> 6)) : 0u;  // cpp:235
                store_state(0, v, forb, g, 0);
                if (t0On && lane == 0) atoms[(long long)M * (1 + R)] = (u64)__double_as_longlong(g);
                if (lane == 0) {  // the pool starts with the root, not yet emitted, not yet split
                    poolG[0] = g;
                    poolS[0] = 0;
                    samp[0] = g;
                    ctrl->n = 1;
                    ctrl->emitted = 0;
                }
            }
        }
        __syncthreads();
        KW_ACC(1, __builtin_readcyclecounter() - tRoot);  // [1] root (incl. barrier)
        if (uni32(ctrl->stop) == 3) {  // infeasible: kBest2D returns 0 (cpp:588-593)
            if (tid == 0) { p.nf[b] = 0; if (p.pushed) p.pushed[b] = 0; }
            continue;
        }

        // ---- phase 1b: the column order of the enumeration (point 8; as in kbest_engine.hip) ----------
        // Columns that are dear to change first, cheap ones last; key = the exact cost of taking a column's row away with nothing
        // else fixed (one search per column from the root's duals; every wave brings the root into its own working set).  The
        // gains are still summed and the tables written in the reference's column order (posOf / colOf).
        // (not under root-subtree sharding: the shards of kbest_c.h are those of the reference's column order, kbest_engine.hip)
        if (prune && p.rootColStride <= 1 && M >= 3 && k >= 3 && !(p.flags & (KBEST_FLAG_EXACT_ROOT | KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_REORDER))) {
            double v[R];
            u32 all = 0;
            {
                const unsigned char *st0 = stBase;
                const double *su = reinterpret_cast<const double *>(st0), *sv = reinterpret_cast<const double *>(st0 + offV);
                const int *sr = reinterpret_cast<const int *>(st0 + offR4C), *sc = reinterpret_cast<const int *>(st0 + offC4R);
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    v[i] = 0.0;
                    if (r < D) { uW[r] = su[r]; v[i] = sv[r]; r4cW[r] = sr[r]; c4rW[r] = sc[r]; all |= 1u << i; }
                }
            }
            wave_fence();
            double *key = childG;  // (M doubles: the children's gain list, empty before round 0)
#pragma unroll 1
            for (int c = wave; c < M; c += NWV) {
                const int fr = uni32(r4cW[c]);
                const u32 frBit = (lane == (fr & 63)) ? (1u << (fr >> 6)) : 0u;
                double spc[R], delta;
                int pred[R], sink = 0;
                u32 scanned;
                const int dj = wide_dijkstra<R, false>(Cw, D, uW, c4rW, lane, v, all, frBit, c, INF, spc, pred, scanned, delta, sink, M, nullptr, fr);
                if (lane == 0) key[c] = (dj || delta != delta) ? INF : delta;  // (never a NaN: the ranks below must be a permutation)
            }
            __syncthreads();
            for (int c = tid; c < M; c += NT) {  // positions by descending key, equal keys by column
                const double kc = key[c];
                int rank = 0;
                for (int j = 0; j < M; j++) {
                    const double kj = key[j];
                    rank += (kj > kc || (kj == kc && j < c)) ? 1 : 0;
                }
                posOf[c] = (unsigned short)rank;
                colOf[rank] = (unsigned short)c;
            }
            __syncthreads();
            if (wave == 0) {  // the root in that order: u and row4col by position, col4row's values are positions, v as it is
                double uN[R];
                int rN[R], cN[R];
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    uN[i] = 0.0; rN[i] = -1; cN[i] = -1;
                    if (r < D) {
                        const int oc = r < M ? (int)colOf[r] : r;
                        uN[i] = uW[oc];
                        rN[i] = r4cW[oc];
                        const int cOld = c4rW[r];
                        cN[i] = (cOld >= 0 && cOld < M) ? (int)posOf[cOld] : cOld;
                    }
                }
                wave_fence();
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    if (r < D) { uW[r] = uN[i]; r4cW[r] = rN[i]; c4rW[r] = cN[i]; }
                }
                wave_fence();
                const int r0 = uni32(r4cW[0]);
                const u32 forb = (lane == (r0 & 63)) ? (1u << (r0 >> 6)) : 0u;  // cpp:235: the row of the FIRST column of the order
                store_state(0, v, forb, poolG[0], 0);
            }
            for (int i = tid; i < N * M; i += NT) {  // the cost copy's real columns again, in the new order (D == N)
                const int c = i / N, r = i - c * N;
                const double x = Cg[r + (long long)colOf[c] * N];
                double val = maximize ? (-x + cdelW) : (x - cdelW);  // cpp:558 / cpp:564
                if (val != val) val = INF;
                Cw[i] = val;
            }
            __syncthreads();
        }

        // ---- phase 2: rounds (kBest2D cpp:607-634 + split cpp:455-532 as a batched frontier) ----
        // The pool holds the candidates that are NOT yet emitted, sorted by gain; an entry = (gain, state slot | SPLIT).
        // Per round the first `spec` not-yet-split entries are split together (speculatively, except the minimum): the
        // pool always holds a partition of the not yet emitted assignments, so the emitted sequence is the reference's
        // (point 5).  With spec = 1 this is the reference's order of operations exactly (that mode
        // counts its pushes).  The head goes out while it has been split; the first entry that has not been split is
        // split in this round: it is emitted too, but ends the run (its children are not in the pool yet).
        const int spec = p.spec;
        for (int round = 0;; round++) {
            // (cheap expressions of the lane number are recomputed where they are used instead of being hoisted out of the
            //  round loop and spilled: kbest_engine.hip, the same remark)
            asm volatile("" : "+v"(lane), "+v"(tid));
            const int cur = uni32(ctrl->cur), n = uni32(ctrl->n), emitted = uni32(ctrl->emitted);
            const double *srcG = poolG + (long long)cur * p.poolStride;
            const int *srcS = poolS + (long long)cur * p.poolStride;
            double *dstG = poolG + (long long)(1 - cur) * p.poolStride;
            int *dstS = poolS + (long long)(1 - cur) * p.poolStride;
            if (t0On && round == 1) {
                double *sd = reinterpret_cast<double *>(nodeBase);
                u64 *sm = reinterpret_cast<u64 *>(nodeBase) + 64;
                int *cnt = reinterpret_cast<int *>(reinterpret_cast<u64 *>(nodeBase) + 64 + 64 * R);
                wide_apriori_threshold<R, NT>(sd, sm, cnt, childG, atoms, M, k, &ctrl->t0);
                __syncthreads();
            }
            // -- select + emission bookkeeping (wave 0)
            KW_T(tSel);
            KW_ACC(2, 1);  // [2] rounds
            if (wave == 0) {
                int cnt = 0, firstU = -1;
                for (int base = 0; base < n && cnt < spec; base += 64) {
                    const int i = base + lane;
                    const bool open = i < n && !(srcS[i] & WIDE_SPLIT);
                    const u64 m = __ballot(open);
                    if (m) {
                        const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
                        if (firstU < 0) firstU = base + __builtin_ctzll(m);
                        if (open && rank < spec) ctrl->selIdx[rank] = i;
                        cnt += __popcll(m);
                    }
                }
                const int nsel = cnt < spec ? cnt : spec;
                int run = (nsel > 0) ? firstU + 1 : n;
                if (run > k - emitted) run = k - emitted;
                if (round == 0 && lane == 0) {
                    const double g0 = srcG[0];
                    ctrl->cutoffGain = maximize ? (g0 - p.cutoff) : (g0 + p.cutoff);          // cpp:681/684
                    ctrl->gain0u = maximize ? (-g0 + ctrl->cdelta) : (g0 + ctrl->cdelta);    // cpp:599-603
                }
                wave_fence();
                const double gain0u = ctrl->gain0u;
                bool cutStop = false;
                int nEmit = run;
                if (useCut) {  // cpp:709-719: the first slot beyond gainBest[0] +- cutoff is written, not counted, and ends the call
                    for (int base = 0; base < run && !cutStop; base += 64) {
                        const int j = base + lane;
                        bool beyond = false;
                        if (j < run) {
                            const double g = srcG[j];
                            const double gu = maximize ? (-g + ctrl->cdelta) : (g + ctrl->cdelta);
                            beyond = maximize ? (gu < gain0u - p.cutoff) : (gu > gain0u + p.cutoff);
                        }
                        const u64 m = __ballot(beyond);
                        if (m) { cutStop = true; nEmit = base + __builtin_ctzll(m); }
                    }
                }
                for (int j = lane; j < nEmit + (cutStop ? 1 : 0); j += 64)
                    if (emitted + j < k) {
                        const double g = srcG[j], gu = maximize ? (-g + ctrl->cdelta) : (g + ctrl->cdelta);  // cpp:626-630
                        if (emitted + j < p.kTab) p.gain[outBase + emitted + j] = gu;
                        else if (j < nEmit) p.tieGain[b] = gu;  // (tie mode only: the counted solution behind the tables)
                    }
                wave_fence();
                if (lane < nsel) {  // the nodes of this round: state slot, gain, active column
                    const int idx = ctrl->selIdx[lane];
                    const int sid = srcS[idx] & WIDE_SID_MASK;
                    ctrl->selSid[lane] = sid;
                    ctrl->selG[lane] = srcG[idx];
                    ctrl->selA[lane] = *reinterpret_cast<const int *>(stBase + (long long)sid * p.stateStride + offTail + 8);
                }
                wave_fence();
                if (lane == 0) {
                    int off = 0;
                    for (int s2 = 0; s2 < nsel; s2++) { ctrl->selOff[s2] = off; off += M - ctrl->selA[s2]; }
                    ctrl->selOff[nsel] = off;
                    ctrl->nsel = nsel;
                    ctrl->nEmit = nEmit;
                    ctrl->lastSel = nsel > 0 ? ctrl->selIdx[nsel - 1] : -1;
                    if (cutStop || emitted + nEmit >= k || nsel == 0) ctrl->stop = 1;
                }
            }
            __syncthreads();
            KW_T(tOut);
            KW_ACC(3, tOut - tSel);  // [3] select + emission bookkeeping (wave 0; the others wait)
            const int nsel = uni32(ctrl->nsel), nEmit = uni32(ctrl->nEmit), lastSel = uni32(ctrl->lastSel);
            // -- outputs of the hypotheses emitted in this round (cpp:618-630): from their saved states
            for (int idx = tid; idx < nEmit * (N + M); idx += NT) {
                const int j = idx / (N + M), q = idx - j * (N + M);
                if (emitted + j >= p.kTab) continue;  // (the solution behind the tables: its gain is all that is kept)
                const unsigned char *E = stBase + (long long)(srcS[j] & WIDE_SID_MASK) * p.stateStride;
                // (the states are in the enumeration's column order: the tables in the reference's)
                if (q < M) put_index(p.row4col, (outBase + emitted + j) * p.ldCol + colOf[q], reinterpret_cast<const int *>(E + offR4C)[q], tabI8);
                else if (p.col4row) {
                    const int cv = reinterpret_cast<const int *>(E + offC4R)[q - M];
                    put_index(p.col4row, (outBase + emitted + j) * p.ldRow + (q - M), (cv >= 0 && cv < M) ? (int)colOf[cv] : cv, tabI8);
                }
            }
            if (uni32(ctrl->stop) != 0) {
                __syncthreads();
                if (tid == 0 && ctrl->stop == 1) ctrl->emitted = emitted + nEmit;
                break;
            }
            // -- the hypotheses being split come into LDS once (coalesced): every child reads its parent from there
            {
                const int words = (int)(offTail + 16) >> 2, total = nsel * words;
                for (int i0 = tid; i0 < total; i0 += 4 * NT) {  // four loads in flight per thread
                    int val[4], s2v[4], wv[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int i = i0 + e * NT < total ? i0 + e * NT : i0;
                        s2v[e] = i / words;
                        wv[e] = i - s2v[e] * words;
                        val[e] = reinterpret_cast<const int *>(stBase + (long long)ctrl->selSid[s2v[e]] * p.stateStride)[wv[e]];
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (i0 + e * NT < total) reinterpret_cast<int *>(nodeBase + (size_t)s2v[e] * L.nodeStride)[wv[e]] = val[e];
                }
            }
            __syncthreads();
            const int eNew = emitted + nEmit;
            const int nOld = n - nEmit, Rk = k - eNew;  // Rk: candidates that can still be output
            double T = (nOld >= Rk) ? srcG[nEmit + Rk - 1] : INF;
            const double cutG = ctrl->cutoffGain;
            if (useCut && !maximize && cutG < T) T = cutG;
            if (ctrl->t0 < T) T = ctrl->t0;  // (+inf until round 1, or when it is off)
            const double cmaxv = ctrl->cmax;

            // -- children of the selected hypotheses (split, cpp:455-532), one wave each, dynamic queue over (node, column).
            //    (A separate first-step pass over all children -- wave w tests the columns a + w, a + w + NW, ... of each node
            //    with one ballot per child -- was measured and is slower: only 6 children in 10 end at their first step, the
            //    others take 2-4, and the extra barrier costs more than the saved state loads: 10.4 -> 11.4 ms on 256 x 128x128.)
            const int totalItems = uni32(ctrl->selOff[nsel]);
            int npush = 0;
            KW_T(tCh);
            KW_ACC(4, tCh - tOut);  // [4] outputs of the emitted hypotheses
            for (;;) {
                KW_T(tA);
                int t = 0;
                if (lane == 0) t = atomicAdd(&ctrl->nextTicket, 1);
                t = uni32(t);
                if (t >= totalItems) break;
                int s2 = 0;
                for (int h = WIDE_MAX_SPEC >> 1; h >= 1; h >>= 1)  // the node of ticket t: last s2 with selOff[s2] <= t
                    if (s2 + h < nsel && t >= ctrl->selOff[s2 + h]) s2 += h;
                s2 = uni32(s2);
                const int a = uni32(ctrl->selA[s2]), ps = uni32(ctrl->selSid[s2]);
                const int c = a + (t - uni32(ctrl->selOff[s2]));
                // (the root's children; the partition is on the REFERENCE's column, kbest_c.h, whatever order the enumeration works in)
                if (round == 0 && p.rootColStride > 1 && ((int)colOf[c < M ? c : 0] % p.rootColStride) != p.rootColOffset) continue;
                const double pgain = ctrl->selG[s2];
                const double bound = (prune && T < INF) ? (T - pgain) + 1e-9 * (fabs(T) + cmaxv) : INF;
                const unsigned char *P = nodeBase + (size_t)s2 * L.nodeStride;  // the parent, in LDS
                const double *Pu = reinterpret_cast<const double *>(P), *Pv = reinterpret_cast<const double *>(P + offV);
                const int *Pr4c = reinterpret_cast<const int *>(P + offR4C), *Pc4r = reinterpret_cast<const int *>(P + offC4R);
                const u32 pforb = reinterpret_cast<const u32 *>(P + offForb)[lane];
                double v[R], spc[R];
                int pred[R];
                u32 cand = 0;
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    v[i] = 0.0;
                    if (r < D) {
                        v[i] = Pv[r];
                        if (Pc4r[r] >= c) cand |= 1u << i;  // rows of columns >= c: cpp:480-488, 525-527
                    }
                }
                const int fr = uni32(Pr4c[c]);  // row freed: cpp:277-278
                const u32 frBit = (lane == (fr & 63)) ? (1u << (fr >> 6)) : 0u;
                const u32 forbm = (c == a) ? pforb : frBit;  // cpp:490 / cpp:510-516
                u32 scanned;
                double delta;
                int sink = 0;
                KW_T(tB);
                KW_ACC(5, tB - tA);  // [5] ticket + parent state load
                KW_ACC(11, 1);       // [11] children started
                // the search runs on the parent's u and col4row (shared, read-only); only a child that completes gets
                // its own copy, to be updated and saved
#ifdef KB_PROFILE
                int nSteps = 0;
                const int dj = wide_dijkstra<R, true>(Cw, D, Pu, Pc4r, lane, v, cand, forbm, c, bound, spc, pred, scanned, delta, sink,
                                                      (p.flags & KBEST_FLAG_EXACT_ROOT) ? 0x7fffffff : M, &nSteps, fr);
                KW_ACC(13, nSteps);  // [13] Dijkstra steps
#else
                const int dj = wide_dijkstra<R, true>(Cw, D, Pu, Pc4r, lane, v, cand, forbm, c, bound, spc, pred, scanned, delta, sink,
                                                      (p.flags & KBEST_FLAG_EXACT_ROOT) ? 0x7fffffff : M, nullptr, fr);
#endif
                KW_T(tC);
                KW_ACC(6, tC - tB);  // [6] shortest augmenting path
                if (dj) continue;
                KW_ACC(12, 1);       // [12] children completed
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int r = lane + 64 * i;
                    if (r < D) { uW[r] = Pu[r]; c4rW[r] = Pc4r[r]; r4cW[r] = Pr4c[r]; }
                }
                wave_fence();
                if (lane == 0) { c4rW[fr] = -1; r4cW[c] = -1; }
                wave_fence();
                wide_update<R>(uW, c4rW, r4cW, predW, lane, v, spc, pred, scanned, delta, sink, c, D);
                KW_T(tD);
                KW_ACC(7, tD - tC);  // [7] dual update + augmentation
                const double g = wide_gain<R>(Cw, D, M, r4cW, lane, posOf);
                KW_T(tE);
                KW_ACC(8, tE - tD);  // [8] gain
                if (useCut && (maximize ? (g < cutG) : (g > cutG))) continue;  // cutHyp, cpp:496/521
                npush++;
                if (t0On && ps == 0) {  // a child of the root: the optimum with one alternating path applied -- cost and rows moved
                    u64 mw[R];
#pragma unroll
                    for (int i = 0; i < R; i++) {
                        const int r = lane + 64 * i;
                        bool moved = false;
                        if (r < D) {
                            const int cOld = Pc4r[r] >= M ? M : Pc4r[r], cNew = c4rW[r] >= M ? M : c4rW[r];  // (zero-padded columns: one place)
                            moved = cOld != cNew;
                        }
                        mw[i] = __ballot(moved);
                    }
                    if (lane == 0) {
                        atoms[(long long)c * (1 + R)] = (u64)__double_as_longlong(g - pgain);
#pragma unroll
                        for (int i = 0; i < R; i++) atoms[(long long)c * (1 + R) + 1 + i] = mw[i];
                    }
                }
                int sid = -1;
                if (lane == 0) {
                    const int idx = atomicSub(&ctrl->nFree, 1) - 1;
                    if (idx >= 0) sid = freeList[idx];
                }
                sid = uni32(sid);
                if (sid < 0) {  // cannot happen: S = k + spec * maxCol + 2 covers pool + children of a round
                    if (lane == 0) ctrl->stop = 2;
                    break;
                }
                const int rn = uni32(r4cW[c]);
                const u32 forbN = forbm | ((lane == (rn & 63)) ? (1u << (rn >> 6)) : 0u);  // cpp:362
                store_state(sid, v, forbN, g, c);
                if (lane == 0) {
                    const int pos = atomicAdd(&ctrl->nChild, 1);
                    childG[pos] = g;
                    childS[pos] = sid;
                    childC[pos] = (ps << 10) | c;  // (parent, column): the order of exact ties
                }
                KW_ACC(9, __builtin_readcyclecounter() - tE);  // [9] state store + push
            }
            KW_T(tW);
            if ((p.flags & KBEST_FLAG_COUNT_PUSHED) && lane == 0 && npush) atomicAdd(&ctrl->pushed, npush);
            __syncthreads();
            KW_T(tM);
            KW_ACC(10, tM - tW);  // [10] wait at the barrier after the children
            if (uni32(ctrl->stop) != 0) break;

            // -- merge: the old entries that were not emitted and the children, sorted, first Rk kept; the entries selected
            //    in this round are split now; the states of the emitted ones (all of them split by now) are released
            const int nChild = uni32(ctrl->nChild);
            for (int j = tid; j < nEmit; j += NT) freeList[atomicAdd(&ctrl->nFree, 1)] = srcS[j] & WIDE_SID_MASK;
            // the children first sort themselves in LDS (rank among the children: gain, then (parent, column)); an old
            // entry then finds how many children go before it by a binary search there -- with k in the thousands
            // (bruteForceProb) the pool is what is long, and a linear pass over the children per entry was the round
            {
                constexpr int EPC = (1024 + NT - 1) / NT < 2 ? 2 : (1024 + NT - 1) / NT;  // wide_lds_layout: at most 1024 children per round
                double cg[EPC];
                int cs[EPC], cr[EPC];
#pragma unroll
                for (int e = 0; e < EPC; e++) {
                    const int j = tid + e * NT;
                    cr[e] = -1;
                    if (j < nChild) {
                        const double g = childG[j];
                        const int cj = childC[j];
                        int r = 0;
                        for (int j2 = 0; j2 < nChild; j2++) {
                            const double g2 = childG[j2];
                            r += (g2 < g || (g2 == g && childC[j2] < cj)) ? 1 : 0;
                        }
                        cg[e] = g; cs[e] = childS[j]; cr[e] = r;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPC; e++)
                    if (cr[e] >= 0) { childG[cr[e]] = cg[e]; childS[cr[e]] = cs[e]; }
                __syncthreads();
            }
            // (four entries per thread and pass: four loads in flight, four searches of the same fixed depth side by side;
            //  the sorted children are padded with +inf to a power of two P > nChild)
            int P = 1;
            while (P <= nChild) P <<= 1;
            for (int j = nChild + tid; j < P; j += NT) childG[j] = INF;
            // where a child goes: the number of pool entries <= its gain.  samp[] holds every 64th gain of the pool (kept
            // up to date by whoever writes the pool): a wave finds the 64-entry block in LDS, then ONE coalesced read of
            // that block and a ballot -- instead of a binary search of ~15 dependent HBM reads per child
            const bool useSamp = (long long)k + 1 <= 64LL * WIDE_SAMPLES;
            if (useSamp) {
                const int ns = (n + 63) >> 6;
                for (int j = wave; j < nChild; j += NWV) {
                    const double g = childG[j];
                    int q = 0;
                    for (int base = 0; base < ns; base += 64) {
                        const u64 m = __ballot(base + lane < ns && samp[base + lane] <= g);
                        q += __popcll(m);
                        if (m != ~0ull) break;
                    }
                    int ub = 0;
                    if (q > 0) {
                        const int idx = (q - 1) * 64 + lane;
                        ub = (q - 1) * 64 + __popcll(__ballot(idx < n && srcG[idx] <= g));
                    }
                    if (lane == 0) childC[j] = ub > nEmit ? ub - nEmit : 0;  // (every emitted entry is <= any new child)
                }
            } else {
                for (int j = tid; j < nChild; j += NT) {
                    const double g = childG[j];
                    int lo = 0, hi = nOld;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (srcG[nEmit + mid] <= g) lo = mid + 1; else hi = mid;
                    }
                    childC[j] = lo;
                }
            }
            __syncthreads();
            for (int i0 = tid; i0 < nOld; i0 += 4 * NT) {
                double g[4];
                int sv[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = i0 + e * NT, j = nEmit + (i < nOld ? i : i0);
                    g[e] = srcG[j];
                    sv[e] = srcS[j] | ((j <= lastSel) ? WIDE_SPLIT : 0);
                    lo[e] = 0;
                }
                for (int h = P >> 1; h >= 1; h >>= 1) {  // children with a smaller gain (ties: pool entries before new children)
#pragma unroll
                    for (int e = 0; e < 4; e++) lo[e] += (childG[lo[e] + h - 1] < g[e]) ? h : 0;
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = i0 + e * NT;
                    if (i < nOld) {
                        const int pos = i + lo[e];
                        if (pos < Rk) {
                            dstG[pos] = g[e];
                            dstS[pos] = sv[e];
                            if (useSamp && (pos & 63) == 0) samp[pos >> 6] = g[e];
                        } else {
                            freeList[atomicAdd(&ctrl->nFree, 1)] = sv[e] & WIDE_SID_MASK;
                        }
                    }
                }
            }
            for (int j = tid; j < nChild; j += NT) {
                const double g = childG[j];
                const int pos = childC[j] + j;
                if (pos < Rk) {
                    dstG[pos] = g;
                    dstS[pos] = childS[j];
                    if (useSamp && (pos & 63) == 0) samp[pos >> 6] = g;
                } else {
                    freeList[atomicAdd(&ctrl->nFree, 1)] = childS[j];
                }
            }
            __syncthreads();
            if (tid == 0) {
                int nNew = nOld + nChild;
                if (nNew > Rk) nNew = Rk;
                ctrl->emitted = eNew;
                ctrl->n = nNew;
                ctrl->cur = 1 - cur;
                ctrl->nChild = 0;
                ctrl->nextTicket = 0;
            }
            __syncthreads();
            KW_ACC(14, __builtin_readcyclecounter() - tM);  // [14] merge
        }
        __syncthreads();
#ifdef KB_PROFILE
        profAcc[15] = __builtin_readcyclecounter() - profT0;  // [15] whole problem (this wave)
        if (p.prof && lane == 0)
            for (int i = 0; i < 16; i++) atomicAdd(p.prof + (long long)b * 16 + i, profAcc[i]);
#endif
        if (tid == 0) {
            p.nf[b] = (ctrl->stop == 2) ? -3 : (ctrl->emitted > p.kTab ? p.kTab : ctrl->emitted);
            if (p.pushed) p.pushed[b] = ctrl->pushed;
        }
    }
    if (p.queue && tid == 0 && atomicAdd(p.queue + 1, 1u) == gridDim.x - 1u) {
        __hip_atomic_store(p.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int R, bool TILE, int NWV>
static hipError_t launch_wide_rtn(const WideParams &p, int grid, hipStream_t stream)
{
    const WideLds L = wide_lds_layout(p.maxRow, p.maxCol, TILE, NWV, p.spec);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_wide_kernel<R, TILE, NWV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((kbest_wide_kernel<R, TILE, NWV>), dim3(grid), dim3(NWV * 64), L.total, stream, p);
    return hipGetLastError();
}

template <int R, bool TILE>
static hipError_t launch_wide_rt(const WideParams &p, int grid, hipStream_t stream)
{
    return p.nw == 16 ? launch_wide_rtn<R, TILE, 16>(p, grid, stream) : launch_wide_rtn<R, TILE, 8>(p, grid, stream);
}

template <int R>
static hipError_t launch_wide_r(const WideParams &p, int grid, hipStream_t stream)
{
    return p.tile ? launch_wide_rt<R, true>(p, grid, stream) : launch_wide_rt<R, false>(p, grid, stream);
}

hipError_t launch_kbest_wide(const WideParams &p, int grid, hipStream_t stream)
{
    if (p.maxRow <= 64) return launch_wide_r<1>(p, grid, stream);
    if (p.maxRow <= 128) return launch_wide_r<2>(p, grid, stream);
    if (p.maxRow <= 256) return launch_wide_rt<4, false>(p, grid, stream);  // 256^2 doubles never fit
    if (p.maxRow <= 512) return launch_wide_rt<8, false>(p, grid, stream);
    return launch_wide_rtn<16, false, 4>(p, grid, stream);  // 513 .. 1 024 rows: 16 rows per lane, four waves per problem
}

}  // namespace kb
