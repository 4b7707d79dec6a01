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
//   * lazy children: a child is solved for its assignment and its exact gain
//     only (no dual update); the queue keeps (gain, parent, column).  The
//     child that is eventually popped is re-solved from its parent's saved
//     state (k-1 extra single augmentations per problem) -- same arithmetic,
//     same result as the reference's eager child;
//   * bounded queue: only the (k - emitted) smallest candidates can ever be
//     output, so the queue is a sorted LDS array of at most k entries, merged
//     by rank each sweep;
//   * early termination: Dijkstra's running distance `delta` is a lower
//     bound of the child's gain (parent gain + delta), so a child is dropped
//     as soon as that bound exceeds the current k-th best candidate (with a
//     safety margin far above rounding error).
//
// Results are identical to the reference for every hypothesis that is output:
// the same assignments in the same order, gains bit-identical (serial
// column-order sum, calcGain cpp:59-80), duals of emitted hypotheses
// bit-identical (updateDualAndAugment cpp:82-117 evaluated in the same order).
//
// fp64 add/sub/compare only -- no MFMA (nothing here is a contraction).
// Compiled WITHOUT fast-math: ((delta + C) - u) - v must not be reassociated.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kbest_engine.h"

namespace kb {

__device__ __forceinline__ double d_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

// ---------------------------------------------------------------- wave tools
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double min_keep(double a, double b) { return b < a ? b : a; }

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// fp64 min over the 64 lanes, returned wave-uniform (set-up code only; the hot
// loop uses the integer-key form below).  All lanes must be active.
__device__ __forceinline__ double wave_min_f64(double x)
{
    x = min_keep(x, dpp_f64<0xB1, 0xF>(x));   // quad_perm [1,0,3,2]
    x = min_keep(x, dpp_f64<0x4E, 0xF>(x));   // quad_perm [2,3,0,1]
    x = min_keep(x, dpp_f64<0x141, 0xF>(x));  // row_half_mirror
    x = min_keep(x, dpp_f64<0x140, 0xF>(x));  // row_mirror
    x = min_keep(x, dpp_f64<0x142, 0xA>(x));  // row_bcast:15 -> rows 1,3
    x = min_keep(x, dpp_f64<0x143, 0xC>(x));  // row_bcast:31 -> rows 2,3
    return readlane_f64(x, 63);
}

// Order-preserving integer key of a double: (khi as int32, klo as uint32)
// compared lexicographically == IEEE '<' on the doubles (no NaNs; -0.0 cannot
// occur in a reduced cost).  Negative values (rounding can make a
// tight arc's reduced cost -1e-17) have their magnitude bits flipped.
__device__ __forceinline__ void to_key(double x, int &khi, u32 &klo)
{
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int s = hi >> 31;
    khi = hi ^ (int)((u32)s >> 1);
    klo = (u32)(lo ^ s);
}
__device__ __forceinline__ double from_key(int khi, u32 klo)  // same involution
{
    const int s = khi >> 31;
    return __hiloint2double(khi ^ (int)((u32)s >> 1), (int)klo ^ s);
}
constexpr int KEY_INF_HI = 0x7ff00000;  // key of +inf is (0x7ff00000, 0)

// One VOP2+DPP instruction per butterfly stage; `s_nop 1` covers the two wait
// states a DPP read needs after a VALU write of the same VGPR (the assembler
// does not pad inline asm).  The result lands in lane 63 and is read into an
// SGPR.  EXEC must be all ones.
#define KB_DPP_MIN_CHAIN(OP)                                                             \
    "s_nop 1\n\t" OP " %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"         \
    "s_nop 1\n\t" OP " %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"              \
    "s_nop 1\n\t" OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" "v_readlane_b32 %0, %1, 63\n\t"

__device__ __forceinline__ int wave_min_i32(int x)
{
    int r;
    asm volatile(KB_DPP_MIN_CHAIN("v_min_i32_dpp") : "=s"(r), "+v"(x));
    return r;
}
__device__ __forceinline__ u32 wave_min_u32(u32 x)
{
    u32 r;
    asm volatile(KB_DPP_MIN_CHAIN("v_min_u32_dpp") : "=s"(r), "+v"(x));
    return r;
}

// force a wave-uniform 64-bit value into SGPRs (values loaded from LDS live in VGPRs)
__device__ __forceinline__ u64 uni64(u64 x)
{
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(x >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ int uni32(int x) { return __builtin_amdgcn_readfirstlane(x); }

// per-lane select driven directly by a 64-bit scalar lane mask
__device__ __forceinline__ int sel32(u64 mask, int ifset, int ifclear)
{
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(ifclear), "v"(ifset), "s"(mask));
    return r;
}

__device__ __forceinline__ u64 bit64(int i) { return 1ull << (i & 63); }

// ------------------------------------------------------------ one augmentation
// Shortest augmenting path from column `start` (lane = row).  Restates the
// do{}while of shortestPathCPP (cpp:168-226) / shortestPathUpdateCPP
// (cpp:297-356):
//   cand    rows still in Row2Scan (bit r)
//   forb    rows skipped while the start column itself is scanned (cpp:310)
//   c4r     this lane's row -> column, -1 = unassigned (a sink)
//   u       LDS array, duals per column; v this lane's row dual
// shortestPathCost[row] is held per lane as an order-preserving integer key
// (to_key): the strict '<' update (cpp:185, 314) is one 64-bit integer compare
// and the arg-min (cpp:191-194, 320-323: first minimum in ascending row order)
// is two 6-stage DPP min chains (high word, then low word among the lanes that
// tie on the high word) + ballot + ff1.  A scanned row's key is overwritten
// with +inf so it never wins again; when FULL, its distance is kept in dv for
// the dual update (a scanned row's shortestPathCost never changes afterwards).
// Returns 0 = path found, 1 = infeasible (cpp:197, 327), 2 = abandoned because
// delta exceeds the bound key (only when EARLY).
template <bool EARLY, bool FULL>
__device__ __forceinline__ int dijkstra(const double *Cs, int LDC, const double *u, int rl, int lane,
                                        double v, int c4r, u64 cand, u64 forb, int start, int bndHi,
                                        u32 bndLo, int &pred, double &dv, u64 &scannedOut,
                                        double &deltaOut, int &sinkOut)
{
    int khi = KEY_INF_HI;
    u32 klo = 0;
    int dvlo = 0, dvhi = 0;
    cand = uni64(cand);
    u64 scanned = 0, act = cand & ~uni64(forb);
    int cur = uni32(start);
    bndHi = uni32(bndHi);
    bndLo = (u32)uni32((int)bndLo);
    double delta = 0.0;
    pred = 0;
    for (int it = 0;; it++) {
        if (it > 64) return 1;  // cannot happen (one row leaves `cand` per step); keeps a bug from hanging the GPU
        const double cval = Cs[rl + cur * LDC];
        const double ucur = u[cur];
        const double rc = ((delta + cval) - ucur) - v;  // cpp:183 / cpp:313, left to right
        int nhi;
        u32 nlo;
        to_key(rc, nhi, nlo);
        const long long nk = (long long)(((u64)(u32)nhi << 32) | nlo);
        const long long ok = (long long)(((u64)(u32)khi << 32) | klo);
        const u64 upd = __ballot(nk < ok) & act;        // strict '<': cpp:185, 314
        khi = sel32(upd, nhi, khi);
        klo = (u32)sel32(upd, (int)nlo, (int)klo);
        pred = sel32(upd, cur, pred);
        const int mhi = wave_min_i32(khi);
        const u64 m1 = __ballot(khi == mhi);
        const u32 t = (u32)sel32(m1, (int)klo, -1);
        const u32 mlo = wave_min_u32(t);
        if (mhi >= KEY_INF_HI) return 1;                 // minimum is +inf: infeasible
        if (EARLY && (mhi > bndHi || (mhi == bndHi && mlo > bndLo))) return 2;
        const u64 eq = __ballot(t == mlo) & m1;
        const int closest = __ffsll((long long)eq) - 1;  // lowest row index: cpp:191, 320
        const u64 cbit = 1ull << closest;
        scanned |= cbit;
        cand &= ~cbit;
        delta = from_key(mhi, mlo);
        khi = sel32(cbit, KEY_INF_HI, khi);   // retire the row (gfx9 v_writelane cannot take two SGPRs)
        klo = (u32)sel32(cbit, 0, (int)klo);
        if (FULL) {
            dvlo = sel32(cbit, __double2loint(delta), dvlo);
            dvhi = sel32(cbit, __double2hiint(delta), dvhi);
        }
        const int cc = __builtin_amdgcn_readlane(c4r, closest);
        if (cc < 0) { sinkOut = closest; break; }
        cur = cc;
        act = cand;
    }
    dv = __hiloint2double(dvhi, dvlo);
    scannedOut = scanned;
    deltaOut = delta;
    return 0;
}

// updateDualAndAugment (cpp:82-117), lane = row for v / c4r and lane = column
// for r4c; u lives in LDS.  spc = this row's shortestPathCost (valid for
// scanned rows).
__device__ __forceinline__ void dual_update_flip(double *u, int lane, double &v, int &c4r, int &r4c,
                                                 double spc, int pred, u64 scanned, double delta,
                                                 int sink, int start)
{
    const bool sc = ((scanned >> lane) & 1ull) != 0;
    if (sc && lane != sink) {          // scanned columns other than start: cpp:96-99
        const int c = c4r;
        u[c] = u[c] + delta - spc;
    }
    if (lane == 0) u[start] = u[start] + delta;  // cpp:92
    if (sc) v = v - delta + spc;                  // cpp:102-106
    int r = sink, c, guard = 0;
    do {                                          // cpp:108-116
        c = __builtin_amdgcn_readlane(pred, r);
        const int nxt = __builtin_amdgcn_readlane(r4c, c);
        c4r = (lane == r) ? c : c4r;
        r4c = (lane == c) ? r : r4c;
        r = nxt;
    } while (c != start && ++guard < 64);
}

// calcGain (cpp:59-80): serial left-to-right fp64 sum over columns.  Columns
// < from are taken from the caller's partial sum acc0 (the parent's prefix:
// a child on column c only changes columns >= c).  If prefixOut != nullptr the
// partial sums before each column are written there (lane = column).
__device__ __forceinline__ double serial_gain(const double *Cs, int LDC, int lane, int r4c, int from,
                                              int M, double acc0, double *prefixOut)
{
    double t = 0.0;
    if (lane >= from && lane < M) t = Cs[r4c + lane * LDC];
    double acc = acc0, mine = 0.0;
    for (int j = from; j < M; j++) {
        if (lane == j) mine = acc;
        acc = acc + readlane_f64(t, j);
    }
    if (prefixOut && lane >= from && lane < M) prefixOut[lane] = mine;
    return acc;
}

// ------------------------------------------------------------------ the kernel
struct Ctrl {
    double gain;        // shifted gain of the current parent
    double cdelta;      // CDelta * numCol (cpp:583)
    double cutoffGain;  // workMem.cutoffGain (cpp:681/684)
    double cmax;        // largest finite shifted cost (scale of the safety margin)
    double gain0u;      // gainBest[0]
    u64 forb;           // parent's accumulated forbidden rows (forbiddenActiveRows)
    int activeCol;
    int nq;    // entries in the current pool buffer
    int cur;   // current pool buffer
    int head;  // 1 if entry 0 of the current buffer has been popped
    int stop;
    int pushed;
};
static_assert(sizeof(Ctrl) <= 96, "Ctrl must fit the LDS slot reserved by lds_layout");

template <int NW>
__global__ void __launch_bounds__(NW * 64) kbest_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NW * 64;
    const double INF = d_inf();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int N = p.nRow ? p.nRow[b] : p.maxRow;
    const int M = p.nCol ? p.nCol[b] : p.maxCol;
    const int k = p.k;
    if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol) {  // undefined in the reference
        if (tid == 0) p.nf[b] = -1;
        return;
    }
    const int D = N, LDC = D | 1;
    const Lds L = lds_layout(p.maxRow, k);
    double *Cs = reinterpret_cast<double *>(smem + L.offC);
    double *pu = reinterpret_cast<double *>(smem + L.offU);
    double *pv = reinterpret_cast<double *>(smem + L.offV);
    double *prefix = reinterpret_cast<double *>(smem + L.offPrefix);
    double *childGain = reinterpret_cast<double *>(smem + L.offChildGain);
    double *PG[2] = {reinterpret_cast<double *>(smem + L.offPoolG[0]),
                     reinterpret_cast<double *>(smem + L.offPoolG[1])};
    u32 *PM[2] = {reinterpret_cast<u32 *>(smem + L.offPoolM[0]), reinterpret_cast<u32 *>(smem + L.offPoolM[1])};
    int *pr4c = reinterpret_cast<int *>(smem + L.offR4C);
    int *pc4r = reinterpret_cast<int *>(smem + L.offC4R);
    Ctrl *ctrl = reinterpret_cast<Ctrl *>(smem + L.offCtrl);

    const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.maxRow * p.maxCol);
    const bool maximize = p.maximize != 0, useCut = p.useCutoff != 0;
    const bool prune = (p.flags & KBEST_FLAG_NO_PRUNE) == 0;
    const int rl = lane < D ? lane : D - 1;
    const u64 allRows = (D >= 64) ? ~0ull : ((1ull << D) - 1ull);

    // ---- phase 0: makeCostMatrixSafe + zero padding (cpp:534-569, 582-585) --
    {
        double mn = INF;  // min of C, or min of -C when maximising (max C = -min(-C), exact)
        for (int c = wave; c < M; c += NW)
            for (int r = lane; r < N; r += 64) {
                double x = Cg[r + (long long)c * N];
                x = maximize ? -x : x;
                mn = min_keep(mn, x);
            }
        mn = wave_min_f64(mn);
        if (lane == 0) childGain[wave] = mn;
        __syncthreads();
        mn = childGain[0];
        for (int w = 1; w < NW; w++) mn = min_keep(mn, childGain[w]);
        const double cdel = maximize ? -mn : mn;
        __syncthreads();
        double cm = 0.0;
        for (int c = wave; c < D; c += NW)
            for (int r = lane; r < N; r += 64) {
                double val = 0.0;
                if (c < M) {
                    const double x = Cg[r + (long long)c * N];
                    val = maximize ? (-x + cdel) : (x - cdel);  // cpp:558 / cpp:564
                    // inf - inf (e.g. an all-inf matrix) gives NaN; every comparison the reference makes with
                    // a NaN reduced cost is false (cpp:185, 314), i.e. the arc behaves exactly like +inf.  The
                    // integer-key compare below needs that made explicit.
                    if (val != val) val = INF;
                    if (val < INF && val > cm) cm = val;
                }
                Cs[r + c * LDC] = val;
            }
        cm = -wave_min_f64(-cm);
        if (lane == 0) childGain[wave] = cm;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; w++) cm = childGain[w] > cm ? childGain[w] : cm;
            ctrl->cmax = cm;
            ctrl->cdelta = cdel * (double)M;  // cpp:583
            ctrl->stop = 0;
            ctrl->pushed = 0;
            ctrl->nq = 0;
            ctrl->cur = 0;
            ctrl->head = 0;
        }
        __syncthreads();
    }

    unsigned char *stBase = p.states + (long long)b * k * p.stateStride;
    const int offTail = (18 * p.maxRow + 7) & ~7;

    // save the hypothesis held by wave 0 (u in LDS) as state `slot`, and emit it
    auto save_and_emit = [&](int slot, double v, int r4c, int c4r, u64 forb, double gain, int activeCol) {
        unsigned char *st = stBase + (long long)slot * p.stateStride;
        double *su = reinterpret_cast<double *>(st);
        if (lane < D) {
            su[lane] = pu[lane];
            su[p.maxRow + lane] = v;
            st[16 * p.maxRow + lane] = (unsigned char)r4c;
            st[17 * p.maxRow + lane] = (unsigned char)c4r;
        }
        if (lane == 0) {
            *reinterpret_cast<u64 *>(st + offTail) = forb;
            *reinterpret_cast<double *>(st + offTail + 8) = gain;
            *reinterpret_cast<int *>(st + offTail + 16) = activeCol;
        }
        const long long o = (long long)b * k + slot;
        if (lane < M) p.row4col[o * p.maxCol + lane] = r4c;
        if (lane < N) p.col4row[o * p.maxRow + lane] = c4r;
        if (lane == 0) p.gain[o] = maximize ? (-gain + ctrl->cdelta) : (gain + ctrl->cdelta);  // cpp:599-603
    };
    // publish the hypothesis held by wave 0 as the parent of the next split
    auto publish_parent = [&](double v, int r4c, int c4r, u64 forb, double gain, int activeCol) {
        if (lane < D) { pv[lane] = v; pr4c[lane] = r4c; pc4r[lane] = c4r; }
        if (lane == 0) { ctrl->gain = gain; ctrl->forb = forb; ctrl->activeCol = activeCol; }
    };

    // ---- phase 1: root LAP (shortestPathCPP, cpp:119-238) on wave 0 ----------
    if (wave == 0) {
        if (lane < D) pu[lane] = 0.0;
        double v = 0.0, spc, delta;
        int c4r = -1, r4c = -1, pred, sink = 0;
        u64 scanned;
        bool bad = false;
        for (int c = 0; c < D; c++) {
            if (dijkstra<false, true>(Cs, LDC, pu, rl, lane, v, c4r, allRows, 0ull, c, KEY_INF_HI, 0u, pred, spc,
                                      scanned, delta, sink)) { bad = true; break; }
            dual_update_flip(pu, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, c);
        }
        if (bad) {
            if (lane == 0) ctrl->stop = 1;
        } else {
            const double g = serial_gain(Cs, LDC, lane, r4c, 0, M, 0.0, prefix);
            const u64 forb = bit64(__builtin_amdgcn_readlane(r4c, 0));  // cpp:235
            publish_parent(v, r4c, c4r, forb, g, 0);
            if (lane == 0) {
                ctrl->cutoffGain = maximize ? (g - p.cutoff) : (g + p.cutoff);          // cpp:681/684
                ctrl->gain0u = maximize ? (-g + ctrl->cdelta) : (g + ctrl->cdelta);
            }
            save_and_emit(0, v, r4c, c4r, forb, g, 0);
        }
    }
    __syncthreads();
    if (uni32(ctrl->stop)) {  // infeasible: kBest2D returns 0 (cpp:588-593)
        if (tid == 0) { p.nf[b] = 0; if (p.pushed) p.pushed[b] = 0; }
        return;
    }

    // ---- phase 2: Murty sweeps (kBest2D loop cpp:607-634, split cpp:455-532) --
    int nf = k;
    for (int s = 0;; s++) {
        if (s + 1 >= k) break;
        // control values come out of LDS in VGPRs: make them provably wave-uniform so that every loop below
        // is a scalar-controlled loop
        const int a = uni32(ctrl->activeCol);
        const int nch = M - a;
        const int R = k - (s + 1);  // candidates that can still be output
        const int src = uni32(ctrl->cur), nqOld = uni32(ctrl->nq), head = uni32(ctrl->head);
        const int nOld = nqOld - head;
        const double pgain = ctrl->gain;
        const double cutG = ctrl->cutoffGain;
        const u64 pforb = uni64(ctrl->forb);
        // early-termination bound on a child's Dijkstra distance: child gain = parent gain + delta (up to
        // rounding), so delta > (T - parent gain) + margin can never enter the k best.  Kept as an integer key.
        int bndHi = KEY_INF_HI;
        u32 bndLo = 0;
        if (prune) {
            double T = (nOld >= R) ? PG[src][head + R - 1] : INF;
            if (useCut && !maximize && cutG < T) T = cutG;
            if (T < INF) to_key((T - pgain) + 1e-9 * (fabs(T) + ctrl->cmax), bndHi, bndLo);
        }
        // -- children of the parent, one wave each (shortestPathUpdateCPP, gain only)
        {
            const double v = (lane < D) ? pv[lane] : 0.0;
            const int c4rP = (lane < D) ? pc4r[lane] : -1;
            const int r4cP = (lane < D) ? pr4c[lane] : -1;
            int npush = 0;
            for (int ci = wave; ci < nch; ci += NW) {
                const int c = a + ci;
                double g = INF;
                const bool skip = (s == 0 && p.rootColStride > 1 && (c % p.rootColStride) != p.rootColOffset);
                if (!skip) {
                    const int fr = __builtin_amdgcn_readlane(r4cP, c);       // row freed: cpp:277-278
                    const u64 cand = __ballot(lane < D && c4rP >= c);         // rows of columns >= c: cpp:480-488, 525-527
                    const u64 forbm = (c == a) ? pforb : bit64(fr);           // cpp:490 / cpp:510-516
                    const int c4r = (lane == fr) ? -1 : c4rP;
                    double spc, delta;
                    int pred, sink = 0;
                    u64 scanned;
                    const int st = dijkstra<true, false>(Cs, LDC, pu, rl, lane, v, c4r, cand, forbm, c, bndHi, bndLo,
                                                         pred, spc, scanned, delta, sink);
                    if (st == 0) {
                        int r4c = (lane == c) ? -1 : r4cP;
                        int r = sink, cc, guard = 0;
                        do {  // path flip, row4col side only (cpp:108-116)
                            cc = __builtin_amdgcn_readlane(pred, r);
                            const int nxt = __builtin_amdgcn_readlane(r4c, cc);
                            r4c = (lane == cc) ? r : r4c;
                            r = nxt;
                        } while (cc != c && ++guard < 64);
                        g = serial_gain(Cs, LDC, lane, r4c, c, M, prefix[c], nullptr);
                        if (useCut && (maximize ? (g < cutG) : (g > cutG))) g = INF;  // cutHyp, cpp:496/521
                        else npush++;
                    }
                }
                if (lane == 0) childGain[ci] = g;
            }
            if ((p.flags & KBEST_FLAG_COUNT_PUSHED) && lane == 0 && npush) atomicAdd(&ctrl->pushed, npush);
        }
        __syncthreads();
        // -- merge the fresh candidates into the sorted pool, keep the R smallest
        const int dst = src ^ 1;
        int nValid = 0;
        for (int j = 0; j < nch; j++) nValid += (childGain[j] < INF) ? 1 : 0;
        nValid = uni32(nValid);
        for (int i = tid; i < nOld; i += NT) {
            const double g = PG[src][head + i];
            int pos = i;
            for (int j = 0; j < nch; j++) pos += (childGain[j] < g) ? 1 : 0;
            if (pos < R) { PG[dst][pos] = g; PM[dst][pos] = PM[src][head + i]; }
        }
        for (int j = tid; j < nch; j += NT) {
            const double g = childGain[j];
            if (g < INF) {
                int lo = 0, hi = nOld;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (PG[src][head + mid] <= g) lo = mid + 1; else hi = mid;
                }
                int pos = lo;
                for (int j2 = 0; j2 < nch; j2++) {
                    const double g2 = childGain[j2];
                    pos += (g2 < g || (g2 == g && j2 < j)) ? 1 : 0;
                }
                if (pos < R) { PG[dst][pos] = g; PM[dst][pos] = ((u32)s << 8) | (u32)(a + j); }
            }
        }
        int nq = nOld + nValid;
        if (nq > R) nq = R;
        __syncthreads();
        if (nq == 0) { nf = s + 1; break; }  // queue empty: cpp:631-633
        // -- pop the best candidate and re-solve it in full from its parent's state
        if (wave == 0) {
            const u32 meta = (u32)uni32((int)PM[dst][0]);
            const int par = (int)(meta >> 8), col = (int)(meta & 255u);
            const unsigned char *st = stBase + (long long)par * p.stateStride;
            const double *su = reinterpret_cast<const double *>(st);
            double v = 0.0;
            int r4c = -1, c4r = -1;
            if (lane < D) {
                pu[lane] = su[lane];
                v = su[p.maxRow + lane];
                r4c = st[16 * p.maxRow + lane];
                c4r = st[17 * p.maxRow + lane];
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
            const int rc = dijkstra<false, true>(Cs, LDC, pu, rl, lane, v, c4r, cand, forbm, col, KEY_INF_HI, 0u, pred,
                                                 spc, scanned, delta, sink);
            if (rc == 0) dual_update_flip(pu, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, col);
            const double g = serial_gain(Cs, LDC, lane, r4c, 0, M, 0.0, prefix);
            const u64 forbN = forbm | bit64(__builtin_amdgcn_readlane(r4c, col));  // cpp:362
            publish_parent(v, r4c, c4r, forbN, g, col);
            save_and_emit(s + 1, v, r4c, c4r, forbN, g, col);
            if (lane == 0) {
                ctrl->cur = dst;
                ctrl->nq = nq;
                ctrl->head = 1;
                if (rc != 0) ctrl->stop = 2;  // cannot happen: the candidate was solved before
                if (useCut) {                 // cpp:709-719
                    const double gu = maximize ? (-g + ctrl->cdelta) : (g + ctrl->cdelta);
                    if (maximize ? (gu < ctrl->gain0u - p.cutoff) : (gu > ctrl->gain0u + p.cutoff)) ctrl->stop = 1;
                }
            }
        }
        __syncthreads();
        const int stop = uni32(ctrl->stop);
        if (stop) { nf = (stop == 2) ? -3 : s + 1; break; }
    }
    if (tid == 0) {
        p.nf[b] = nf;
        if (p.pushed) p.pushed[b] = ctrl->pushed;
    }
}

// ------------------------------------------------- association weights epilogue
// assignmentProb accumulate / normalise (assignment.cpp:616-648) and its
// single-column fast path (assignment.cpp:554-570).  One wave per problem,
// lane = measurement (column); solutions are accumulated in ascending order
// exactly as the reference loop does, so only exp() itself can differ.
__global__ void __launch_bounds__(64) weights_kernel(WeightParams p)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nL = p.nL[b], nM = p.nM[b];
    double *probs = p.probs + p.probOff[b];
    const double GATE = 42.0;  // assignment.cpp:9
    if (nM == 1) {
        const double *cost = p.cost + p.costOff[b];
        if (lane == 0) {
            double norm = 0.0;
            for (int i = 0; i <= nL; i++) {
                double q = 0.0;
                if (cost[i] < GATE) { q = exp(-cost[i]); norm += q; }
                probs[i] = q;
            }
            norm = 1.0 / norm;
            for (int i = 0; i <= nL; i++) probs[i] = probs[i] * norm;
        }
        return;
    }
    const int nf = p.nf[b];
    const double *gain = p.gain + (long long)b * p.k;
    const int *r4c = p.row4col + (long long)b * p.k * p.maxCol;
    for (int i = lane; i < nM * (nL + 1); i += 64) probs[i] = 0.0;
    __syncthreads();
    const double best = gain[0];
    double total = 0.0;
    for (int s = 0; s < nf; s++) {
        const double g = gain[s];
        if (!(best + GATE > g)) continue;  // :622-626
        const double w = exp(best - g);
        total += w;
        if (lane < nM) {
            const int r = r4c[(long long)s * p.maxCol + lane];
            probs[lane * (nL + 1) + (r >= nL ? nL : r)] += w;  // :633-638
        }
    }
    __syncthreads();
    const double norm = 1.0 / total;  // :643
    for (int i = lane; i < nM * (nL + 1); i += 64) probs[i] *= norm;
}

// ------------------------------------------------------------------- launchers
template <int NW>
static hipError_t launch_nw(const Params &p, int B, hipStream_t stream)
{
    const Lds L = lds_layout(p.maxRow, p.k);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_kernel<NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kbest_kernel<NW>, dim3(B), dim3(NW * 64), L.total, stream, p);
    return hipGetLastError();
}

hipError_t launch_kbest(const Params &p, int B, int nWaves, hipStream_t stream)
{
    switch (nWaves) {
    case 1: return launch_nw<1>(p, B, stream);
    case 2: return launch_nw<2>(p, B, stream);
    case 8: return launch_nw<8>(p, B, stream);
    default: return launch_nw<4>(p, B, stream);
    }
}

hipError_t launch_weights(const WeightParams &p, int B, hipStream_t stream)
{
    hipLaunchKernelGGL(weights_kernel, dim3(B), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace kb
