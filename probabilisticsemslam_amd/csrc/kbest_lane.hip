// kbest_lane.hip -- MI355X (gfx950) k-best assignment kernel for problems of up to 32 rows: ONE LANE PER CHILD.
//
// The 64-row kernel (kbest_engine.hip) gives every child of Murty's partition (split, shortestPathCPP.cpp:455-532) a
// wavefront with lane = row: on a 32-row problem half of every wave idles through every Dijkstra step and every DPP
// stage, on a 16-row problem three quarters of it.  Here the mapping is turned around: a child's rows are an unrolled
// loop over registers (shortestPathCost[DR] and pred[DR] per lane, DR = 16 or 32) and the 64 lanes of a wave are 64
// CHILDREN that advance one Dijkstra step per pass (shortestPathUpdateCPP, cpp:297-356):
//
//   * one pass = for every row r: t = ((delta + C[r,cur]) - u[cur]) - v[r] (cpp:313, same order of operations),
//     strict '<' update of shortestPathCost / pred (cpp:314-317), strict '<' running minimum in ascending row order
//     (cpp:320-323: the lowest row among equal minima).  Plain fp64 compares -- no keys, no cross-lane traffic at all.
//     The cost column comes from the LDS tile at a per-lane address (odd column stride: lanes on different columns hit
//     different banks), v[r] from the child's parent in LDS (node blocks are 8 bytes apart modulo the bank row, so
//     lanes whose parents differ do not collide either), u[cur] likewise;
//   * row sets (Row2Scan, cpp:486, 506-508) are one 32-bit mask per lane; rows forbidden for the start column
//     (cpp:310) are masked in the first pass only;
//   * early termination as in the 64-row kernel: the settled distance delta only grows and parent gain + delta is
//     the child's gain, so a lane gives its child up as soon as delta exceeds (T - parent gain) + margin, T = the
//     gain of the pool's last entry.  There is no separate first-step filter: a child that
//     dies at its first step costs one pass of one lane;
//   * lanes whose child is given up (or infeasible, cpp:327) draw the next child of the round from a shared queue
//     as soon as enough of them are idle; lanes whose child reached its sink keep their registers ("parked") until
//     the wave has nothing left to step, then all parked children are finished together, still one per lane:
//     path flip (cpp:108-116) through a per-lane LDS scratch, the reference's serial gain (calcGain, cpp:59-80),
//     dual update (cpp:92-106) straight into the child's saved state in HBM;
//   * everything around the children -- cost shift, root LAP on one wave with lane = row (shared with the 64-row
//     kernel: kbest_lap.h), sorted LDS pool merged in place by rank, batched frontier (`spec` hypotheses split per
//     round), emission bookkeeping, outputs widened from the saved states at the end -- follows the 64-row kernel;
//     every completed child is kept in full, its state slot comes from an LDS free list and goes back when the
//     candidate drops out of the pool.
//
// Same arithmetic in the same order per emitted hypothesis as the reference, hence the same bits (gains from the
// serial column-order sum, duals by the reference's update).  fp64 add / sub / compare only -- no MFMA.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "kbest_engine.h"
#include "kbest_wave.h"
#include "kbest_lap.h"

namespace kb {

namespace {

constexpr u32 LN_SPLIT = 0x80000000u;  // pool meta: children already generated and merged

struct LaneCtrl {
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
    int nextItem;  // work queue of the children phase
    int nFresh;    // completed children appended this round
    int freeTop;   // free state slots on the stack
    int nComp;     // children that reached their sink this round (compList)
    short selIdx[LANE_MAX_SPEC];           // pool index of each node selected in the last A phase (split in the next round)
    unsigned short selSid[LANE_MAX_SPEC];  // and its state slot
};
static_assert(sizeof(LaneCtrl) <= 160, "LaneCtrl must fit the LDS slot reserved by lane_lds_layout");

__device__ __forceinline__ int lane_rank(u64 mask)  // set bits of `mask` below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}


#ifndef KB_LANE_REFILL_MIN
#define KB_LANE_REFILL_MIN 3
#endif
constexpr int LANE_REFILL_MIN = KB_LANE_REFILL_MIN;  // idle children slots that make a refill of the wave worth its set-up

// One row of one Dijkstra pass for the 64 children of a wave (shortestPathUpdateCPP's scan loop, cpp:307-325, with lane =
// child): t = ((delta + C[r,cur]) - u[cur]) - v[r] (cpp:313); where row R is in the lane's row set (bit R of pm) and
// t < shortestPathCost[r] (strict, cpp:314): take it and remember the column (pred); then the running minimum in
// ascending row order (strict '<', cpp:320: the lowest row among equal minima).  Written by hand because the lane masks
// want to live in EXEC: the compiler's version keeps every row's compare results in SGPR pairs across the whole unrolled
// pass (280 scalar spills at DR = 32).  11 vector + 2 scalar instructions per row; EXEC is put back before the block ends.
template <int R>
__device__ __forceinline__ void lane_row(double &spc, u32 &pred, double &best, int &arg, double Crc, double vr,
                                         double delta, double ucur, u32 pm, u32 curRep, u64 execAll)
{
    double t;
    u32 tmp;
    u64 m;
    const u32 bm = 0xffu << (8 * (R & 3));
    asm volatile(
        "v_and_b32_e32 %[tmp], %[bit], %[pm]\n\t"
        "v_add_f64 %[t], %[delta], %[C]\n\t"
        "v_add_f64 %[t], %[t], -%[u]\n\t"
        "v_add_f64 %[t], %[t], -%[v]\n\t"
        "v_cmpx_ne_u32_e64 %[m], 0, %[tmp]\n\t"
        "v_cmpx_lt_f64_e32 vcc, %[t], %[spc]\n\t"
        "v_mov_b64 %[spc], %[t]\n\t"
        "v_bfi_b32 %[pred], %[bm], %[cr], %[pred]\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_cmpx_lt_f64_e32 vcc, %[spc], %[best]\n\t"
        "v_mov_b64 %[best], %[spc]\n\t"
        "v_mov_b32_e32 %[arg], %[r]\n\t"
        "s_mov_b64 exec, %[ex]\n\t"
        : [spc] "+v"(spc), [pred] "+v"(pred), [best] "+v"(best), [arg] "+v"(arg), [t] "=&v"(t), [tmp] "=&v"(tmp), [m] "=&s"(m)
        : [C] "v"(Crc), [v] "v"(vr), [delta] "v"(delta), [u] "v"(ucur), [pm] "v"(pm), [cr] "v"(curRep), [ex] "s"(execAll),
          [bit] "n"(1u << R), [bm] "s"(bm), [r] "n"(R)
        : "vcc");
}

}  // namespace

// RL: rows per lane (unrolled); G: lanes per child (DR = RL * G rows at most: 16 / 32); NW: waves per problem; EPT: pool entries
// per thread in the in-place merge
template <int RL, int G, int NW, int EPT>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) kbest_lane_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NW * 64, DR = RL * G, LOG_RL = (RL == 16 ? 4 : (RL == 8 ? 3 : 2)), CPW = 64 / G;  // CPW: children per wave
    static_assert((RL == 4 || RL == 8 || RL == 16) && (G == 2 || G == 4) && (DR == 16 || DR == 32), "row / lane split");
    // node block / saved state offsets
    constexpr int N_V = 8 * DR, N_R4C = 16 * DR, N_C4R = 17 * DR, N_CAND = 18 * DR, N_GAIN = 22 * DR, N_BOUND = 22 * DR + 8,
                  N_FORB = 22 * DR + 16, N_A = 22 * DR + 20, N_SID = 22 * DR + 24;
    constexpr int S_R4C = 16 * DR, S_C4R = 17 * DR, S_TAIL = 18 * DR;
    const double INF = d_inf();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int N = p.nRow ? p.nRow[b] : p.maxRow;
    const int M = p.nCol ? p.nCol[b] : p.maxCol;
    const int k = p.k;
    // exact ties (kbest_ties.h): the tables hold p.kTab slots (k, or k - 1: the k-th solution is enumerated for its gain only)
    if (p.tieGain && tid == 0) p.tieGain[b] = __longlong_as_double(0x7ff8000000000000LL);  // no solution behind the tables (yet)
    if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol) {  // undefined in the reference
        if (tid == 0) p.nf[b] = (M == 0 || N == 0) ? 0 : -1;            // (an empty frame: nothing to assign, nothing found)
        return;
    }
    const int D = N, LDC = D | 1;
    const int spec = p.spec;
    const LaneLds L = lane_lds_layout(p.maxRow, p.maxCol, k, p.spec, NW, G);
    double *Cs = reinterpret_cast<double *>(smem + L.offC);
    double *freshG = reinterpret_cast<double *>(smem + L.offFreshG);
    u32 *freshM = reinterpret_cast<u32 *>(smem + L.offFreshM);
    unsigned short *freshS = reinterpret_cast<unsigned short *>(smem + L.offFreshS);
    double *PG = reinterpret_cast<double *>(smem + L.offPoolG);
    u32 *PM = reinterpret_cast<u32 *>(smem + L.offPoolM);
    unsigned short *PS = reinterpret_cast<unsigned short *>(smem + L.offPoolS);
    unsigned short *items = reinterpret_cast<unsigned short *>(smem + L.offItems);
    unsigned short *freeList = reinterpret_cast<unsigned short *>(smem + L.offFree);
    LaneCtrl *ctrl = reinterpret_cast<LaneCtrl *>(smem + L.offCtrl);
    double *gainW = reinterpret_cast<double *>(smem + L.offGainW);
    double *red = freshG;  // cross-wave reduction scratch of phase 0
    unsigned short *slotSid = p.slotSid + (long long)blockIdx.x * slot_table_stride(k);

    const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.ldRow * p.ldCol);
    const bool maximize = p.maximize != 0, useCut = p.useCutoff != 0;
    const bool prune = (p.flags & KBEST_FLAG_NO_PRUNE) == 0;
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const int rl = lane < D ? lane : D - 1;
    const u64 allRows = (D >= 64) ? ~0ull : ((1ull << D) - 1ull);
    const int nSlots = p.statesPerProblem;
    unsigned char *stBase = p.states + (long long)b * nSlots * p.stateStride;
    const long long outBase = (long long)b * p.kTab;
#ifdef KB_PROFILE
    unsigned long long profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long profT0 = __builtin_readcyclecounter();
#endif

    // ---- phase 0: makeCostMatrixSafe + zero padding (cpp:534-569, 582-585) --
    double cdelTile = 0.0;  // the tile's shift (phase 1b loads the columns again in another order)
    {
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
        const double cdel = maximize ? -mn : mn;
        cdelTile = cdel;
        __syncthreads();
        double cm = 0.0;
        for (int c = wave; c < D; c += NW)
            for (int r = lane; r < N; r += 64) {
                double val = 0.0;
                if (c < M) {
                    const double x = Cg[r + (long long)c * N];
                    val = maximize ? (-x + cdel) : (x - cdel);  // cpp:558 / cpp:564
                    if (val != val) val = INF;                   // inf - inf: behaves like +inf in every compare of the reference
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
            ctrl->nFresh = 0;
            ctrl->nComp = 0;
            ctrl->freeTop = nSlots - 1;
            ctrl->selIdx[0] = -1;
            ctrl->selSid[0] = 0;
        }
        for (int i = tid; i < nSlots - 1; i += NT) freeList[i] = (unsigned short)(nSlots - 1 - i);  // slot 0 is the root's
        __syncthreads();
    }

    // Write a full hypothesis (lane = row / column) to state slot `sid`.
    auto store_state = [&](int sid, double u, double v, int r4c, int c4r, u64 forb, double gain, int activeCol) {
        unsigned char *st = stBase + (long long)sid * p.stateStride;
        double *sd = reinterpret_cast<double *>(st);
        if (lane < D) {
            sd[lane] = u;
            sd[DR + lane] = v;
            st[S_R4C + lane] = (unsigned char)r4c;
            st[S_C4R + lane] = (unsigned char)c4r;
        }
        if (lane == 0) {
            *reinterpret_cast<u64 *>(st + S_TAIL) = forb;
            *reinterpret_cast<double *>(st + S_TAIL + 8) = gain;
            *reinterpret_cast<int *>(st + S_TAIL + 16) = activeCol;
        }
    };
    // One wave publishes a hypothesis (lane = row / column) as node block `nbOff`; cand[c] = rows of the columns >= c
    // (the rows a child on column c may scan: cpp:480-488, 525-527) by a suffix OR over the lanes.
    auto fill_node = [&](int nbOff, double u, double v, int r4c, int c4r, u32 forb, double gain, int activeCol, int sid) {
        u32 m = (lane < D) ? (1u << (r4c & 31)) : 0u;
#pragma unroll
        for (int s = 1; s < DR; s <<= 1) {
            const u32 o = (u32)__shfl_down((int)m, s);
            m |= (lane + s < 64) ? o : 0u;
        }
        if (lane < D) {
            *reinterpret_cast<double *>(smem + nbOff + 8 * lane) = u;
            *reinterpret_cast<double *>(smem + nbOff + N_V + 8 * lane) = v;
            smem[nbOff + N_R4C + lane] = (unsigned char)r4c;
            smem[nbOff + N_C4R + lane] = (unsigned char)c4r;
            *reinterpret_cast<u32 *>(smem + nbOff + N_CAND + 4 * lane) = m;
        }
        if (lane == 0) {
            *reinterpret_cast<double *>(smem + nbOff + N_GAIN) = gain;
            *reinterpret_cast<u32 *>(smem + nbOff + N_FORB) = forb;
            *reinterpret_cast<int *>(smem + nbOff + N_A) = activeCol;
            *reinterpret_cast<int *>(smem + nbOff + N_SID) = sid;
        }
    };

    // ---- phase 1: root LAP (shortestPathCPP, cpp:119-238) on wave 0, lane = row -> node 0, state 0, slot 0 ----
    if (wave == 0) {
        double *uR = reinterpret_cast<double *>(smem + L.offNodes);  // node block 0's u
        if (lane < D) uR[lane] = 0.0;
        double v = 0.0, spc, delta;
        int c4r = -1, r4c = -1, pred, sink = 0;
        u64 scanned;
        bool bad = false;
        u64 todo = allRows;  // columns still to be augmented from
        {
            // Column reduction first (Jonker-Volgenant's initialisation), as in the 64-row kernel: u[c] = min of column c,
            // a row that is the arg-min of some column goes to the lowest such column; the augmentations then run only
            // from the columns left over.  Same optimal assignment (unique for tie-free costs), another optimal dual pair.
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
            const bool can = lane < M && m < INF;  // (only the REAL columns take part)
            if (can) atomicMin(&owner[am], lane);
            wave_fence();
            if (lane < D) uR[lane] = can ? m : 0.0;
            r4c = (can && owner[am] == lane) ? am : -1;
            const int ow = owner[lane];
            c4r = (lane < D && ow < 64) ? ow : -1;
            wave_fence();
            if (M == D) {
                // Row reduction on top (square problems), as in the 64-row kernel: a row without a column takes
                // v[r] = min over c of (C[r,c] - u[c]) and, if the column of that minimum is free, that column (the lowest row
                // wins): a third fewer augmentations and half their Dijkstra steps.
                double m2 = INF;
                int a2 = 0;
                const int rr = lane < D ? lane : D - 1;
                for (int c0 = 0; c0 < D; c0 += 4) {
                    double x[4], uu[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int cj = (c0 + i < D) ? c0 + i : D - 1;
                        x[i] = Cs[rr + cj * LDC];
                        uu[i] = uR[cj];
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
            if (dijkstra<false>(Cs, LDC, uR, rl, lane, v, c4r, allRows, 0ull, c, INF, spc, pred, scanned, delta, sink)) { bad = true; break; }
            dual_update_flip(uR, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, c);
        }
        if (!bad && M < D) {
            // The zero-padded columns (cpp:582-585): padded column M + j <- the j-th free row, u = 0 (kbest_engine.hip)
            const u64 freeRows = __ballot(lane < D && c4r < 0);
            if (lane < D && c4r < 0) c4r = M + __popcll(freeRows & ((1ull << lane) - 1ull));
            int *slot = reinterpret_cast<int *>(gainW);
            wave_fence();
            if (lane < D && c4r >= M) slot[c4r - M] = lane;
            wave_fence();
            if (lane >= M && lane < D) { r4c = slot[lane - M]; uR[lane] = 0.0; }
            wave_fence();
        }
        if (bad) {
            if (lane == 0) ctrl->stop = 3;
        } else {
            const double g = serial_gain(Cs, LDC, lane, r4c, M, gainW);
            const u32 forb = 1u << (__builtin_amdgcn_readlane(r4c, 0) & 31);  // cpp:235
            const double uMine = (lane < D) ? uR[lane] : 0.0;
            wave_fence();
            store_state(0, uMine, v, r4c, c4r, (u64)forb, g, 0);
            fill_node(L.offNodes, uMine, v, r4c, c4r, forb, g, 0, 0);
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
        if (tid == 0) { p.nf[b] = 0; if (p.pushed) p.pushed[b] = 0; }
        return;
    }
    // ---- phase 1b: the column order of the enumeration (as in the 64-row kernel, kbest_engine.hip) --------------------
    // The columns that are dear to change go first, the cheap ones last: the children that carry the k best then have nearly
    // everything fixed.  Key of column c: the exact cost of taking its row away with nothing else fixed -- one search from the
    // root's duals per column, the waves share them.  Gains are still summed in the reference's column order and the tables
    // written in it.  colOf[position] = the reference's column, posOf = its inverse (in the root's scratch line, free now).
    unsigned char *colOf = reinterpret_cast<unsigned char *>(gainW), *posOf = colOf + 64;
    if (tid < 64) { colOf[tid] = (unsigned char)tid; posOf[tid] = (unsigned char)tid; }
    __syncthreads();
    // (not under root-subtree sharding: the shards of kbest_c.h are those of the reference's column order, kbest_engine.hip)
    const bool reorder = prune && p.rootColStride <= 1 && M >= 3 && k >= 3 && !(p.flags & (KBEST_FLAG_COUNT_PUSHED | KBEST_FLAG_NO_REORDER));
    if (reorder) {
        // (the keys: wave 0's scratch line behind the two index tables; the closure's matrix: the lists of the rounds, all unused
        //  before round 0)
        double *key = gainW + 16;
        const int nb0 = L.offNodes;
        double *u0 = reinterpret_cast<double *>(smem + nb0);
        const int Dp = (D + 15) & ~15;
        if (Dp * Dp * 4 <= L.offFree - L.offFreshG) {
            // all keys from ONE blocked all-pairs closure of the column graph (column_keys_closure, kbest_lap.h): for 16 columns a
            // single 16 x 16 block on one wave -- 16 lock-step passes instead of 16 searches shared by the waves
            column_keys_closure<NW>(reinterpret_cast<float *>(smem + L.offFreshG), key, Cs, LDC, u0,
                                    reinterpret_cast<const double *>(smem + nb0 + N_V), smem + nb0 + N_R4C, D, M);
        } else {
            const double v0 = (lane < D) ? *reinterpret_cast<const double *>(smem + nb0 + N_V + 8 * lane) : 0.0;
            const int c4r0 = (lane < D) ? (int)smem[nb0 + N_C4R + lane] : -1;
            const int r4c0 = (lane < D) ? (int)smem[nb0 + N_R4C + lane] : -1;
#pragma unroll 1
            for (int c = wave; c < M; c += NW) {
                const int fr = __builtin_amdgcn_readlane(r4c0, c);
                const int c4r = (lane == fr) ? -1 : c4r0;
                double spc, delta;
                int pred, sink = 0;
                u64 scanned;
                const int st = dijkstra<false>(Cs, LDC, u0, rl, lane, v0, c4r, allRows, 1ull << fr, c, INF, spc, pred, scanned, delta, sink);
                if (lane == 0) key[c] = (st == 0) ? delta : INF;
            }
        }
        __syncthreads();
        if (wave == 0) {
            double kl = key[lane < M ? lane : 0];
            kl = (kl == kl) ? kl : INF;  // (a NaN key would break the ranks' uniqueness: any order is valid, a non-permutation is not)
            int rank = 0;
            for (int j = 0; j < M; j++) {  // positions by descending key, equal keys by column
                const double kj = readlane_f64(kl, j);
                rank += (kj > kl || (kj == kl && j < lane)) ? 1 : 0;
            }
            if (lane < M) { posOf[lane] = (unsigned char)rank; colOf[rank] = (unsigned char)lane; }
            wave_fence();
            // the root in that order: u and row4col by position, col4row's values are positions, v as it is
            const int oc = (lane < M) ? (int)colOf[lane] : lane;
            const double uN = (lane < D) ? u0[oc] : 0.0;
            const int r4cN = (lane < D) ? (int)smem[nb0 + N_R4C + oc] : 0;
            const int cOld = (lane < D) ? (int)smem[nb0 + N_C4R + lane] : 0;
            const int c4rN = (cOld < M) ? (int)posOf[cOld] : cOld;
            const double vN = (lane < D) ? *reinterpret_cast<const double *>(smem + nb0 + N_V + 8 * lane) : 0.0;
            const double g = *reinterpret_cast<const double *>(smem + nb0 + N_GAIN);
            wave_fence();
            const u32 forb = 1u << (__builtin_amdgcn_readlane(r4cN, 0) & 31);  // cpp:235: the row of the FIRST column of the order
            store_state(0, uN, vN, r4cN, c4rN, (u64)forb, g, 0);
            fill_node(nb0, uN, vN, r4cN, c4rN, forb, g, 0, 0);
        }
        __syncthreads();
        for (int c = wave; c < M; c += NW) {  // the tile's real columns again, in the new order
            const int oc = colOf[c];
            for (int r = lane; r < N; r += 64) {
                const double x = Cg[r + (long long)oc * N];
                double val = maximize ? (-x + cdelTile) : (x - cdelTile);  // cpp:558 / cpp:564
                if (val != val) val = INF;
                Cs[r + c * LDC] = val;
            }
        }
        __syncthreads();
    }
    KB_ACC(0, __builtin_readcyclecounter() - profT0);  // [0] set-up + root solve

    const int scrBase = L.offScr + wave * CPW * L.scrStride;
    const int part = lane & (G - 1), partRow = part * RL;  // this lane's share of a child's rows
    constexpr int QP_UP1 = 1 | (1 << 2) | (3 << 4) | (3 << 6);  // quad_perm [1,1,3,3]: lane <- lane + 1
    constexpr int QP_UP2 = 2 | (3 << 2) | (2 << 4) | (3 << 6);  // quad_perm [2,3,2,3]: lane <- lane + 2
    constexpr int QP_BC = (G == 4) ? 0 : (2 << 4) | (2 << 6);    // quad_perm [0,0,0,0] / [0,0,2,2]: the group's first lane
    unsigned short *compList = reinterpret_cast<unsigned short *>(smem + L.offComp);
    const int ldc8 = LDC * 8;
    // ---- phase 2: rounds ----------------------------------------------------------------------------
    while (uni32(ctrl->stop) == 0) {
        KB_T(tRound);
        KB_ACC(7, 1);  // [7] rounds
        const int nsel = uni32(ctrl->nsel);
        const int emitted = uni32(ctrl->emitted);
        const int R = k - emitted;  // candidates that can still be output
        const int nqEnd = uni32(ctrl->nq), head = uni32(ctrl->head);
        const int nOld = nqEnd - head;
        const double cutG = ctrl->cutoffGain;
        // the candidates selected in the last A phase are split in this round: flag them in the pool now (nobody reads
        // the pool's meta words before the merge, which is behind the barrier after the children phase)
        if (wave == 0 && lane < nsel) {
            const int idx = ctrl->selIdx[lane];
            if (idx >= 0) PM[idx] |= LN_SPLIT;
        }
        // threshold of the pool: once it holds R candidates only children below its largest can matter
        double T = (nOld >= R) ? PG[head + R - 1] : INF;
        if (useCut && !maximize && cutG < T) T = cutG;
        const double cmaxv = ctrl->cmax;
        // Per node: the early-termination bound on the Dijkstra distance (child gain = parent gain + delta up to rounding, so
        // delta > (T - parent gain) + margin can never enter the k best), and the round's list of children (node, column).
        // Every wave writes both itself -- identical values -- and reads only what it has written: no barrier.
        // (lane w < nsel fetches node w's active column and gain at once; the loop over the nodes then runs on register reads,
        //  not on a chain of dependent LDS reads)
        int totalItems = 0;
        {
            const int nbL = L.offNodes + (lane < nsel ? lane : 0) * L.nodeStride;
            const int aL = *reinterpret_cast<const int *>(smem + nbL + N_A);
            if (lane < nsel) {
                const double pgain = *reinterpret_cast<const double *>(smem + nbL + N_GAIN);
                *reinterpret_cast<double *>(smem + nbL + N_BOUND) = (prune && T < INF) ? (T - pgain) + 1e-9 * (fabs(T) + cmaxv) : INF;
            }
            for (int w = 0; w < nsel; w++) {
                const int a = __builtin_amdgcn_readlane(aL, w);
                if (a + lane < M) items[totalItems + lane] = (unsigned short)((w << 6) | (a + lane));
                totalItems += (M - a) > 0 ? (M - a) : 0;
            }
        }
        wave_fence();
        KB_T(tB0);
        KB_ACC(14, tB0 - tRound);  // [14] round prologue

        // -- B: the children, G lanes each -------------------------------------------------------------------------
        {
            double spc[RL];       // shortestPathCost of this lane's rows part * RL .. part * RL + RL - 1
            u32 predw[RL / 4];    // pred of those rows, one byte each
#pragma unroll
            for (int r = 0; r < RL; r++) spc[r] = INF;
#pragma unroll
            for (int q = 0; q < RL / 4; q++) predw[q] = 0u;
            int nb = L.offNodes, cst = 0, fr = 0, cur = 0;   // (uniform over the G lanes of a child)
            u32 cm = 0u, pm = 0u;                             // this lane's rows still in Row2Scan / scanned in this pass (RL bits)
            double delta = 0.0, bound = INF;
            u64 activeM = 0ull;  // lanes stepping a child (wave-uniform mask, whole groups)
            bool more = totalItems > 0;
            const u64 execAll = __builtin_amdgcn_read_exec();
            for (;;) {
                // refill: idle lane groups draw children from the round's queue -- when there are enough of them to be worth
                // the set-up, or when nothing else is left to step
                const u64 idleM = ~activeM;
                const int nIdle = __popcll(idleM) / G;
                if (more && nIdle > 0 && (activeM == 0ull || nIdle >= LANE_REFILL_MIN)) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ctrl->nextItem, nIdle);
                    base = uni32(base);
                    const int avail = totalItems - base;
                    if (avail < nIdle) more = false;
                    const bool idle = __builtin_amdgcn_inverse_ballot_w64(idleM);
                    const int rank = lane_rank(idleM) / G;  // idle groups below this one
                    const bool got = idle && rank < avail;
                    const u64 gotM = __ballot(got);
                    if (gotM) {
                        KB_ACC(4, __popcll(gotM) / G);  // [4] children started
                        const int e = got ? (int)items[base + rank] : 0;
                        const int w = e >> 6, c = e & 63;
                        const int nbN = L.offNodes + w * L.nodeStride;
                        const int frN = smem[nbN + N_R4C + c];                                        // row freed: cpp:277-278
                        const u32 cand = *reinterpret_cast<const u32 *>(smem + nbN + N_CAND + 4 * c);  // rows of columns >= c
                        const int a = *reinterpret_cast<const int *>(smem + nbN + N_A);
                        const u32 nforb = *reinterpret_cast<const u32 *>(smem + nbN + N_FORB);
                        const int sidP = *reinterpret_cast<const int *>(smem + nbN + N_SID);
                        const u32 forb = (c == a) ? nforb : (1u << frN);                               // cpp:490 / cpp:510-516
                        // (root-subtree sharding partitions on the REFERENCE's column, kbest_c.h: the enumeration's own order depends on the launch)
                        bool live = got;
                        if (__builtin_expect(p.rootColStride > 1, 0)) {  // (the stride opaque: its reciprocal must not be kept across the rounds)
                            int st = p.rootColStride;
                            asm volatile("" : "+s"(st));
                            live = got && !(sidP == 0 && ((int)colOf[c] % st) != p.rootColOffset);
                        }
                        const double bnd = *reinterpret_cast<const double *>(smem + nbN + N_BOUND);
                        const u32 candP = (cand >> partRow) & ((1u << RL) - 1u), forbP = (forb >> partRow) & ((1u << RL) - 1u);
                        nb = got ? nbN : nb;
                        cst = got ? c : cst;
                        fr = got ? frN : fr;
                        cur = got ? c : cur;
                        cm = got ? candP : cm;
                        pm = got ? (live ? (candP & ~forbP) : 0u) : pm;
                        delta = got ? 0.0 : delta;
                        bound = got ? bnd : bound;
#pragma unroll
                        for (int r = 0; r < RL; r++) spc[r] = got ? INF : spc[r];
                        activeM |= __ballot(live);
                    }
                }
                if (activeM == 0ull) break;  // (the queue is empty too: a refill was tried)
                KB_ACC(5, __popcll(activeM) / G);  // [5] child Dijkstra steps
                KB_ACC(9, 1);                      // [9] passes
                // one Dijkstra step of every active child (cpp:307-325): every lane scans its RL rows ...
                const int colOff = L.offC + cur * ldc8 + partRow * 8;
                const int vOff = nb + N_V + partRow * 8;
                const double ucur = *reinterpret_cast<const double *>(smem + nb + 8 * cur);
                const u32 curRep = (u32)cur * 0x01010101u;
                double best = INF;
                int arg = 0;
                {
                    double Crc[RL], vr[RL];
#pragma unroll
                    for (int i = 0; i < RL; i++) {
                        Crc[i] = *reinterpret_cast<const double *>(smem + colOff + 8 * i);
                        vr[i] = *reinterpret_cast<const double *>(smem + vOff + 8 * i);
                    }
#define LN_ROW(R) lane_row<(R)>(spc[R], predw[(R) >> 2], best, arg, Crc[R], vr[R], delta, ucur, pm, curRep, execAll)
                    LN_ROW(0); LN_ROW(1); LN_ROW(2); LN_ROW(3);
                    if constexpr (RL > 4) { LN_ROW(RL > 4 ? 4 : 0); LN_ROW(RL > 4 ? 5 : 0); LN_ROW(RL > 4 ? 6 : 0); LN_ROW(RL > 4 ? 7 : 0); }
                    if constexpr (RL > 8) {
                        LN_ROW(RL > 8 ? 8 : 0); LN_ROW(RL > 8 ? 9 : 0); LN_ROW(RL > 8 ? 10 : 0); LN_ROW(RL > 8 ? 11 : 0);
                        LN_ROW(RL > 8 ? 12 : 0); LN_ROW(RL > 8 ? 13 : 0); LN_ROW(RL > 8 ? 14 : 0); LN_ROW(RL > 8 ? 15 : 0);
                    }
#undef LN_ROW
                }
                // ... and the G lanes of a child combine their minima towards the group's first lane: the partner always
                // holds HIGHER rows, so a strict '<' keeps the lowest row among equal minima (cpp:320); then all take its result
                arg += partRow;
                {
                    {
                        const double ob = dpp_f64<QP_UP1, 0xF>(best);
                        const int oa = __builtin_amdgcn_update_dpp(arg, arg, QP_UP1, 0xF, 0xF, false);
                        const bool take = ob < best;
                        best = take ? ob : best;
                        arg = take ? oa : arg;
                    }
                    if constexpr (G == 4) {
                        const double ob = dpp_f64<QP_UP2, 0xF>(best);
                        const int oa = __builtin_amdgcn_update_dpp(arg, arg, QP_UP2, 0xF, 0xF, false);
                        const bool take = ob < best;
                        best = take ? ob : best;
                        arg = take ? oa : arg;
                    }
                    best = dpp_f64<QP_BC, 0xF>(best);
                    arg = __builtin_amdgcn_update_dpp(arg, arg, QP_BC, 0xF, 0xF, false);
                }
                const bool act = __builtin_amdgcn_inverse_ballot_w64(activeM);
                const bool dead = !(best < INF) || (best > bound);  // infeasible (cpp:327) / beyond the k best
                const bool sink = act && !dead && arg == fr;        // the only unassigned row of a child problem (cpp:325)
                const bool cont = act && !dead && arg != fr;
                delta = act ? best : delta;
                const u32 mybit = ((arg >> LOG_RL) == part) ? (1u << (arg & (RL - 1))) : 0u;
                cm = act ? (cm & ~mybit) : cm;
                pm = cont ? cm : 0u;
                const int nxt = smem[nb + N_C4R + arg];
                cur = cont ? nxt : cur;
                activeM = __ballot(cont);
                const u64 sinkM = __ballot(sink);
                if (sinkM) {
                    // A child reached its sink.  What its search leaves behind -- shortestPathCost and pred per row, the rows
                    // still unscanned, the final distance -- goes into the state slot the finished hypothesis will occupy,
                    // and the lanes are free again at once; all completed children of the round are finished together below.
                    int slot = 0;
                    if (sink && part == 0) {
                        const int top = atomicSub(&ctrl->freeTop, 1);
                        slot = top >= 1 ? (int)freeList[top - 1] : -1;
                    }
                    slot = __builtin_amdgcn_update_dpp(slot, slot, QP_BC, 0xF, 0xF, false);  // the group's first lane's
                    if (__ballot(sink && slot < 0)) {  // cannot happen (lane_states_per_problem): fail loudly, not silently
                        if (lane == 0) ctrl->stop = 2;
                    }
                    if (sink && slot >= 0) {
                        unsigned char *st = stBase + (long long)slot * p.stateStride;
                        double *sd = reinterpret_cast<double *>(st);
#pragma unroll
                        for (int i = 0; i < RL; i++) sd[DR + partRow + i] = spc[i];
#pragma unroll
                        for (int q = 0; q < RL / 4; q++) *reinterpret_cast<u32 *>(st + S_R4C + partRow + 4 * q) = predw[q];
                        *reinterpret_cast<unsigned short *>(st + S_C4R + 2 * part) = (unsigned short)cm;
                        if (part == 0) {
                            *reinterpret_cast<u32 *>(st + S_TAIL) = (u32)cst | ((u32)fr << 8) | ((u32)nb << 16);
                            *reinterpret_cast<double *>(st + S_TAIL + 8) = delta;
                            const int pos = atomicAdd(&ctrl->nComp, 1);
                            compList[pos] = (unsigned short)slot;
                        }
                    }
                }
            }
        }
        KB_T(tFin0);
        KB_ACC(1, tFin0 - tB0);  // [1] stepping busy (this wave)
        __syncthreads();           // (every wave's stores above are complete and visible to the workgroup)
        KB_T(tFin1);
        KB_ACC(11, tFin1 - tFin0);    // [11] wait at the barrier after stepping
        // -- B': finish the completed children (shortestPathUpdateCPP's tail, cpp:340-364), 64 / G at a time per wave, G lanes each
        {
            const int nComp = uni32(ctrl->nComp);
            int npush = 0;
            for (int c0 = wave * CPW; c0 < nComp; c0 += NW * CPW) {
                const int ci = c0 + lane / G;
                const bool mine = ci < nComp;
                const int slot = mine ? (int)compList[ci] : 0;
                unsigned char *st = stBase + (long long)slot * p.stateStride;
                double *sd = reinterpret_cast<double *>(st);
                double spc[RL];
                u32 predw[RL / 4];
#pragma unroll
                for (int i = 0; i < RL; i++) spc[i] = sd[DR + partRow + i];
#pragma unroll
                for (int q = 0; q < RL / 4; q++) predw[q] = *reinterpret_cast<const u32 *>(st + S_R4C + partRow + 4 * q);
                const u32 cm = *reinterpret_cast<const unsigned short *>(st + S_C4R + 2 * part);
                const u32 ids = *reinterpret_cast<const u32 *>(st + S_TAIL);
                const double delta = *reinterpret_cast<const double *>(st + S_TAIL + 8);
                const int cst = ids & 255, fr = (ids >> 8) & 255, nb = mine ? (int)(ids >> 16) : L.offNodes;
                const int sb = scrBase + (lane / G) * L.scrStride;  // [0, DR) row4col  [DR, 2DR) col4row  [2DR, 4DR) (pred, its row)
                if (mine) {
                    // scratch: the parent's maps (to be flipped along the path) and, per row, (pred, the row that holds pred's
                    // column in the parent): the walk below then needs ONE dependent read per step
#pragma unroll
                    for (int q = 0; q < RL / 4; q++) {
                        const int o4 = partRow + 4 * q;
                        *reinterpret_cast<u32 *>(smem + sb + o4) = *reinterpret_cast<const u32 *>(smem + nb + N_R4C + o4);
                        *reinterpret_cast<u32 *>(smem + sb + DR + o4) = *reinterpret_cast<const u32 *>(smem + nb + N_C4R + o4);
                    }
#pragma unroll
                    for (int i = 0; i < RL; i++) {
                        const u32 pc = (predw[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        const u32 pr = smem[nb + N_R4C + (pc < (u32)D ? pc : 0u)];
                        *reinterpret_cast<unsigned short *>(smem + sb + 2 * DR + 2 * (partRow + i)) = (unsigned short)(pc | (pr << 8));
                    }
                }
                wave_fence();
                {   // path flip, sink -> start (cpp:108-116), by the first lane of each group
                    int r = fr, guard = 0;
                    bool go = mine && part == 0;
                    while (__ballot(go)) {
                        if (go) {
                            const int e = *reinterpret_cast<const unsigned short *>(smem + sb + 2 * DR + 2 * r);
                            const int cc = e & 255;
                            smem[sb + DR + r] = (unsigned char)cc;
                            smem[sb + cc] = (unsigned char)r;
                            r = e >> 8;
                            go = (cc != cst) && (++guard < 64);
                        }
                    }
                }
                wave_fence();
                // calcGain (cpp:59-80): serial left-to-right sum over the M real columns, from 0.0 (every lane of the group, the
                // same broadcast reads).  The rows first, then the terms, then the chain of adds: no dependent LDS round trips.
                double g = 0.0;
#pragma unroll
                for (int j0 = 0; j0 < DR; j0 += 8) {
                    if (j0 < M) {
                        int rows[8];
                        double term[8];
                        int posj[8];  // position of the reference's column j0 + i in the enumeration's order
#pragma unroll
                        for (int i = 0; i < 8; i++) posj[i] = (j0 + i < M) ? (int)posOf[j0 + i] : 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) rows[i] = (mine && j0 + i < M) ? (int)smem[sb + posj[i]] : 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) term[i] = *reinterpret_cast<const double *>(smem + L.offC + (rows[i] + posj[i] * LDC) * 8);
#pragma unroll
                        for (int i = 0; i < 8; i++) g = (j0 + i < M) ? g + term[i] : g;
                    }
                }
                const bool keep = mine && !(useCut && (maximize ? (g < cutG) : (g > cutG)));  // cutHyp, cpp:496/521
                npush += __popcll(__ballot(keep)) / G;
                KB_ACC(6, __popcll(__ballot(keep)) / G);  // [6] children completed
                if (mine && !keep && part == 0) freeList[atomicAdd(&ctrl->freeTop, 1)] = (unsigned short)slot;  // beyond the cutoff: the slot goes back
                if (keep) {
                    const u32 cand0 = (*reinterpret_cast<const u32 *>(smem + nb + N_CAND + 4 * cst) >> partRow) & ((1u << RL) - 1u);
                    const u32 scanned = cand0 & ~cm;
                    // dual update (cpp:92-106) by rows: row r's column in the parent is c4rP[r] (the freed row's is the start
                    // column itself), every column has exactly one row
#pragma unroll
                    for (int i = 0; i < RL; i++) {
                        const int r = partRow + i;
                        if (r < D) {
                            const bool sc = ((scanned >> i) & 1u) != 0u;
                            const double vr = *reinterpret_cast<const double *>(smem + nb + N_V + 8 * r);
                            sd[DR + r] = sc ? (vr - delta + spc[i]) : vr;               // cpp:102-106
                            const int colr = smem[nb + N_C4R + r];
                            const double uc = *reinterpret_cast<const double *>(smem + nb + 8 * colr);
                            double uN = uc;
                            if (sc) uN = uc + delta - spc[i];                             // cpp:96-99
                            if (r == fr) uN = uc + delta;                                 // cpp:92 (colr == start)
                            sd[colr] = uN;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < RL / 4; q++) {
                        const int o4 = partRow + 4 * q;
                        *reinterpret_cast<u32 *>(st + S_R4C + o4) = *reinterpret_cast<const u32 *>(smem + sb + o4);
                        *reinterpret_cast<u32 *>(st + S_C4R + o4) = *reinterpret_cast<const u32 *>(smem + sb + DR + o4);
                    }
                    if (part == 0) {
                        const u32 nforb = *reinterpret_cast<const u32 *>(smem + nb + N_FORB);
                        const int a = *reinterpret_cast<const int *>(smem + nb + N_A);
                        const u32 forbm = (cst == a) ? nforb : (1u << fr);
                        const u32 forbN = forbm | (1u << smem[sb + cst]);                    // cpp:362
                        *reinterpret_cast<u64 *>(st + S_TAIL) = (u64)forbN;
                        *reinterpret_cast<double *>(st + S_TAIL + 8) = g;
                        *reinterpret_cast<int *>(st + S_TAIL + 16) = cst;
                        const int sidP = *reinterpret_cast<const int *>(smem + nb + N_SID);
                        const int pos = atomicAdd(&ctrl->nFresh, 1);
                        freshG[pos] = g;
                        freshM[pos] = ((u32)sidP << 8) | (u32)cst;  // (parent state, column) of this candidate
                        freshS[pos] = (unsigned short)slot;
                    }
                }
                wave_fence();
            }
            if ((p.flags & KBEST_FLAG_COUNT_PUSHED) && lane == 0 && npush) atomicAdd(&ctrl->pushed, npush);
        }
        KB_T(tB1);
        KB_ACC(10, tB1 - tFin1);  // [10] finish of completed children
        __syncthreads();
        KB_T(tC0);
        KB_ACC(15, tC0 - tB1);    // [15] wait at the barrier after the finishing pass
        // -- C: rank-merge the fresh candidates into the sorted pool IN PLACE (every thread first pulls its entries into
        //    registers), keep the R smallest.  Ties in gain are ordered by (parent, column), so the result does not depend
        //    on the arrival order of the fresh list.  State slots of entries that drop out go back to the free list.
        const int nFresh = uni32(ctrl->nFresh);
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
                {
                    const double2 *f2 = reinterpret_cast<const double2 *>(freshG);
                    int j = 0;
                    for (; j + 4 <= nFresh; j += 4) {
                        const double2 a = f2[j >> 1], bb = f2[(j >> 1) + 1];
                        pos += ((a.x < g) ? 1 : 0) + ((a.y < g) ? 1 : 0) + ((bb.x < g) ? 1 : 0) + ((bb.y < g) ? 1 : 0);
                    }
                    for (; j < nFresh; j++) pos += (freshG[j] < g) ? 1 : 0;
                }
                og[e] = g;
                om[e] = PM[head + i];
                os[e] = PS[head + i];
                opos[e] = pos;
            }
        }
        // fresh entries: the position of each goes through LDS (the round's child list is dead by now), so that a thread
        // holds nothing across the barrier however many entries it has
        unsigned short *fposA = items;
        for (int i = tid; i < nFresh; i += NT) {
            const double g = freshG[i];
            const u32 mj = freshM[i];
            int lo = 0, hi = nOld;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (PG[head + mid] <= g) lo = mid + 1; else hi = mid;
            }
            int pos = lo, nEq = 0;
            {
                const double2 *f2 = reinterpret_cast<const double2 *>(freshG);
                int j2 = 0;
                for (; j2 + 4 <= nFresh; j2 += 4) {
                    const double2 a = f2[j2 >> 1], bb = f2[(j2 >> 1) + 1];
                    pos += ((a.x < g) ? 1 : 0) + ((a.y < g) ? 1 : 0) + ((bb.x < g) ? 1 : 0) + ((bb.y < g) ? 1 : 0);
                    nEq += ((a.x == g) ? 1 : 0) + ((a.y == g) ? 1 : 0) + ((bb.x == g) ? 1 : 0) + ((bb.y == g) ? 1 : 0);
                }
                for (; j2 < nFresh; j2++) {
                    const double g2 = freshG[j2];
                    pos += (g2 < g) ? 1 : 0;
                    nEq += (g2 == g) ? 1 : 0;
                }
            }
            if (__builtin_expect(nEq > 1, 0))  // another fresh candidate with the same gain: order by (parent, column)
                for (int j3 = 0; j3 < nFresh; j3++) pos += (freshG[j3] == g && freshM[j3] < mj) ? 1 : 0;
            fposA[i] = (unsigned short)pos;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (opos[e] >= 0) {
                if (opos[e] < R) { PG[opos[e]] = og[e]; PM[opos[e]] = om[e]; PS[opos[e]] = os[e]; }
                else freeList[atomicAdd(&ctrl->freeTop, 1)] = os[e];
            }
        for (int i = tid; i < nFresh; i += NT) {
            const int pos = fposA[i];
            if (pos < R) { PG[pos] = freshG[i]; PM[pos] = freshM[i]; PS[pos] = freshS[i]; }
            else freeList[atomicAdd(&ctrl->freeTop, 1)] = freshS[i];
        }
        int nq = nOld + nFresh;
        if (nq > R) nq = R;
        __syncthreads();
        KB_T(tA0);
        KB_ACC(2, tA0 - tC0);     // [2] merge incl. its barrier
        // -- A + D: every wave finds the candidates to split next itself (the pool is stable now): the first `spec` entries
        //    that have not been split yet.  Wave w brings the w-th, (w + NW)-th, ... of them into the node blocks from their
        //    saved states.  Lane 0 of wave 0 also does the emission bookkeeping, which depends only on the pool order and flags.
        int nselNew = 0;
        for (int base = 0; base < nq && nselNew < spec; base += 64) {
            const int i = base + lane;
            const bool open = i < nq && !(PM[i] & LN_SPLIT);
            const int ps = (i < nq) ? (int)PS[i] : 0;
            const u64 m = __ballot(open);
            // (every wave writes the same values and reads back only what it has written itself)
            const int rank = nselNew + __popcll(m & ((1ull << lane) - 1ull));
            if (open && rank < spec) { ctrl->selIdx[rank] = (short)i; ctrl->selSid[rank] = (unsigned short)ps; }
            nselNew += __popcll(m);
        }
        nselNew = nselNew < spec ? nselNew : spec;
        wave_fence();
        if (wave == 0) {
            // emission (kBest2D cpp:607-634), all lanes of wave 0, 64 pool entries per pass: the head goes out while it has
            // been split; the first entry that has NOT been split is the first one selected in this round (selection is in
            // pool order): it is emitted too but ends the run, because its children are not in the pool yet.
            int e = emitted, h = 0, stop = 0;
            const double cdel = ctrl->cdelta, g0u = ctrl->gain0u;
            const int sid0 = (int)ctrl->selSid[0];
            bool more = true;
            for (int base = 0; more && base < nq && e < k; base += 64) {
                const int i = base + lane;
                const bool valid = i < nq;
                const double g = valid ? PG[i] : 0.0;
                const u32 meta = valid ? PM[i] : 0u;
                const int psid = valid ? (int)PS[i] : 0;
                const double gu = maximize ? (-g + cdel) : (g + cdel);  // cpp:626-630
                const bool cutB = useCut && valid && (maximize ? (gu < g0u - p.cutoff) : (gu > g0u + p.cutoff));
                const u64 plainM = __ballot(valid && (meta & LN_SPLIT) && !cutB);
                int run = (~plainM == 0ull) ? 64 : __builtin_ctzll(~plainM);  // leading entries that simply go out
                if (run > k - e) run = k - e;
                if (lane < run) {
                    if (e + lane < p.kTab) p.gain[outBase + e + lane] = gu;
                    else p.tieGain[b] = gu;  // (tie mode only: the solution behind the tables)
                    slotSid[e + lane] = (unsigned short)psid;
                }
                e += run;
                h += run;
                if (run == 64) continue;  // the whole pass went out: next 64 entries
                more = false;
                if (e >= k || base + run >= nq) break;
                // the entry that ended the run
                const bool tSplit = (__ballot((meta & LN_SPLIT) != 0) >> run) & 1ull, tCut = (__ballot(cutB) >> run) & 1ull;
                if (!tSplit && nselNew == 0) break;  // not split and not selected this round: wait
                if (lane == run) {
                    if (e < p.kTab) p.gain[outBase + e] = gu;
                    else if (!tCut) p.tieGain[b] = gu;  // (beyond the cutoff: written in the reference, never counted)
                    slotSid[e] = (unsigned short)(tSplit ? psid : sid0);
                }
                if (tCut) { stop = 1; break; }  // cpp:709-719: slot written, not counted
                e++;
                h++;
            }
            if (e >= k) stop = 1;
            if (h >= nq && nselNew == 0) stop = 1;  // queue empty, nothing left to split: cpp:631-633
            if (lane == 0) {
                ctrl->emitted = e;
                ctrl->nsel = nselNew;
                ctrl->nextItem = 0;
                ctrl->nFresh = 0;
                ctrl->nComp = 0;
                ctrl->nq = nq;
                ctrl->head = h;
                if (stop && ctrl->stop == 0) ctrl->stop = 1;
            }
        }
        for (int j0 = wave; j0 < nselNew; j0 += 4 * NW) {  // four saved states in flight per wave
            double ldU[4], ldV[4], ldGain[4];
            int ldR[4], ldC[4], ldA[4], sidJ[4];
            u64 ldForb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = j0 + i * NW;
                sidJ[i] = (j < nselNew) ? uni32((int)ctrl->selSid[j]) : 0;
                const unsigned char *st = stBase + (long long)sidJ[i] * p.stateStride;
                const double *sd = reinterpret_cast<const double *>(st);
                ldU[i] = 0.0; ldV[i] = 0.0; ldR[i] = 0; ldC[i] = 0;
                if (j < nselNew && lane < D) {
                    ldU[i] = sd[lane];
                    ldV[i] = sd[DR + lane];
                    ldR[i] = st[S_R4C + lane];
                    ldC[i] = st[S_C4R + lane];
                }
                ldForb[i] = *reinterpret_cast<const u64 *>(st + S_TAIL);
                ldGain[i] = *reinterpret_cast<const double *>(st + S_TAIL + 8);
                ldA[i] = *reinterpret_cast<const int *>(st + S_TAIL + 16);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = j0 + i * NW;
                if (j < nselNew) fill_node(L.offNodes + j * L.nodeStride, ldU[i], ldV[i], ldR[i], ldC[i], (u32)ldForb[i], ldGain[i], ldA[i], sidJ[i]);
            }
        }
        KB_T(tA1);
        KB_ACC(3, tA1 - tA0);     // [3] select / emit / node loads busy
        __syncthreads();
        KB_ACC(12, __builtin_readcyclecounter() - tA1);  // [12] wait at the barrier after A
    }
    const int stopCode = uni32(ctrl->stop);
    const int nfAll = (stopCode == 2) ? -3 : uni32(ctrl->emitted);
    const int nf = nfAll > p.kTab ? p.kTab : nfAll;
    // ---- phase 3: outputs.  Slot s holds hypothesis slotSid[s]: widen its saved row4col / col4row --------
    for (int idx = tid; idx < nf * (N + M); idx += NT) {
        const int s = idx / (N + M), j = idx - s * (N + M);
        const unsigned char *st = stBase + (long long)slotSid[s] * p.stateStride;
        // (the states are in the enumeration's column order: the tables in the reference's)
        if (j < M) put_index(p.row4col, (outBase + s) * p.ldCol + colOf[j], st[S_R4C + j], tabI8);
        else if (p.col4row) {
            const int cv = st[S_C4R + (j - M)];
            put_index(p.col4row, (outBase + s) * p.ldRow + (j - M), cv < M ? (int)colOf[cv] : cv, tabI8);
        }
    }
    if (tid == 0) {
        p.nf[b] = nf;
        if (p.pushed) p.pushed[b] = ctrl->pushed;
    }
#ifdef KB_PROFILE
    profAcc[13] = __builtin_readcyclecounter() - profT0;  // [13] whole kernel (this wave)
    if (p.prof && lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(p.prof + (long long)b * 16 + i, profAcc[i]);
#endif
}

// ------------------------------------------------------------------- launcher
template <int RL, int G, int NW, int EPT>
static hipError_t launch_lane_t(const Params &p, int B, hipStream_t stream)
{
    const LaneLds L = lane_lds_layout(p.maxRow, p.maxCol, p.k, p.spec, NW, G);
    static std::atomic<int> granted{0};
    if (L.total > granted.load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_lane_kernel<RL, G, NW, EPT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
        if (e != hipSuccess) return e;
        granted.store(L.total, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((kbest_lane_kernel<RL, G, NW, EPT>), dim3(B), dim3(NW * 64), L.total, stream, p);
    return hipGetLastError();
}

template <int RL, int G, int NW>
static hipError_t launch_lane_nw(const Params &p, int B, hipStream_t stream)
{
    return (p.k <= NW * 64) ? launch_lane_t<RL, G, NW, 1>(p, B, stream) : launch_lane_t<RL, G, NW, 4>(p, B, stream);
}

template <int RL, int G>
static hipError_t launch_lane_rg(const Params &p, int B, int nWaves, hipStream_t stream)
{
    switch (nWaves) {
    case 1: return launch_lane_nw<RL, G, 1>(p, B, stream);
    case 2: return launch_lane_nw<RL, G, 2>(p, B, stream);
    case 4: return launch_lane_nw<RL, G, 4>(p, B, stream);
    default: return hipErrorInvalidValue;
    }
}

// lanesPerChild: 4 (16 children per wave; the rows of a child are split evenly over them).  The form with 2 lanes per child (32
// children per wave, 8 / 16 rows per lane) was built in round 3 and removed in round 6: 214 VGPRs at 32 rows (two waves per
// SIMD), 264 - 588 bytes of scratch per lane in its instantiations, and slower everywhere it was measured (4 096 x 32x32, k = 200:
// 8.3 ms against 5.4; NOTES section 8, round 3) -- the passes of a round are set by its longest child, not by the child slots.
hipError_t launch_kbest_lane(const Params &p, int B, int nWaves, int lanesPerChild, hipStream_t stream)
{
    if (p.maxRow > LANE_MAX_DIM || p.spec < 1 || p.spec > LANE_MAX_SPEC || p.k > 4 * nWaves * 64 || lanesPerChild != 4) return hipErrorInvalidValue;
    if (lane_rows(p.maxRow) == 16) return launch_lane_rg<4, 4>(p, B, nWaves, stream);
    return launch_lane_rg<8, 4>(p, B, nWaves, stream);
}

}  // namespace kb
