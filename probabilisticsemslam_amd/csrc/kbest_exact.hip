// kbest_exact.hip -- MI355X (gfx950): kBest2D / kBest2DCutoff in the REFERENCE's OWN ORDER OF OPERATIONS, for any size.
//
// Every other kernel of this engine restructures Murty's enumeration (batched frontier, bounded pool, a column order of its own,
// implicit zero columns) and proves that what it emits is what the reference emits.  Two things those kernels cannot give:
//   * the reference's order of hypotheses with EXACTLY equal gains -- an artefact of std::priority_queue's binary heap
//     (shortestPathCPP.cpp:30-42, 574): which of two equal gains is popped first depends on the position every element has
//     reached through every push and pop before.  The engine's default is a rule of its own ((gain, row4col) lexicographic,
//     kbest_c.h); a caller who wants the reference's answer on integer-like costs, slot for slot, asks for it with
//     KBEST_FLAG_REFERENCE_ORDER;
//   * problems of more than KBEST_MAX_DIM_WIDE (1 024) rows -- the reference has no size limit (cpp:571-644).
// This kernel is the reference's algorithm as it stands: the zero-padded N x N formulation (cpp:582-585), the root by N
// augmentations in column order (shortestPathCPP, cpp:119-238), one priority queue of FULLY SOLVED hypotheses with libstdc++'s
// own sift rules (bits/stl_heap.h: __push_heap / __adjust_heap), per sweep pop -> split into the children of columns activeCol ..
// numCol-1 in that order (split, cpp:455-532; shortestPathUpdateCPP, cpp:240-365) -> emit the new top (cpp:607-634, 709-719).
// Nothing is pruned, nothing reordered: the heap holds what the reference's heap holds, in the same array positions, so the
// pop order of equal gains is the reference's, and col4row names the padded column every left-over row sits on exactly as the
// reference does (SURVEY 8(a) quirks 6 and 7 do not apply to this kernel).
//
// Two forms of the same sequence of operations:
//   * up to 64 rows (kbest_exact64_kernel): the hypothesis in REGISTERS -- lane = row for v / col4row, lane = column for u /
//     row4col, the forbidden rows a 64-bit mask --, the padded cost copy in LDS, the Dijkstra step the hand-written loop of the LDS
//     kernels (kbest_lap.h), a child a few register copies of its parent; one wave per problem, or EIGHT: the children of a sweep are
//     independent of one another (each is solved completely from the parent), only their pushes have an order -- the waves solve
//     them side by side, wave 0 then pushes them in column order;
//   * 65 .. 1 024 rows (kbest_exactN_kernel: eight waves per problem -- the root's Dijkstra steps shared by all of them, a sweep's
//     children side by side on as many of them as the LDS holds scratch for) and
//     beyond (kbest_exact_kernel: one wave): the lanes of a wave share the rows of a Dijkstra step (row = lane, lane + 64, ...: the
//     reduced costs ((delta + C) - u) - v left to right, the strict-'<' update, the minimum with the LOWEST row among equal values:
//     cpp:183-191, 313-320), the dual update and the copies of a hypothesis; the path flip is one lane's; up to 1 024 rows every
//     wave's scratch of a search and hypothesis-being-solved and the hypothesis being split are in LDS, the cost copy in an HBM
//     work space.
// In all of them the queue is one lane's, in HBM (or LDS where it fits), and the pool of hypotheses (25 N bytes each, one per push: a
// child only reaches it if it is feasible and not cut) is in HBM.  This is the slow, total, literal path -- 256 problems of 64 x 64,
// k = 200 take 15 ms here (1 024 of them 34 ms, one alone 9.7 ms; the reference on one host core: 10 ms per problem) against 0.7 ms
// on the LDS kernel; 1 000 integer-cost 28 x 10 problems 9 ms against 4.5; 64 problems of 200 x 150, k = 50: 108 ms (one: 71 ms);
// two of 1 000 x 12, k = 10: 0.76 s, most of it the root's 1 000 augmentations, whose steps all eight waves share (the compiled
// reference on a host core: 0.71 s for one) -- and is only taken when asked for
// (KBEST_FLAG_REFERENCE_ORDER; the tied problems of a KBEST_FLAG_REFERENCE_TIES call) or when no other kernel takes the size.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kbest_engine.h"
#include "kbest_lap.h"
#include "kbest_wave.h"

namespace kb {

namespace {

constexpr int EXACT_LDS_ROWS = 1024;  // up to here a search's scratch (19 bytes per row), the hypothesis being solved and the one being split (25 each) lie in LDS

struct ExLayout {  // byte offsets inside one slot of the work space (D = maxRow, H = hypotheses per slot)
    long long C, spc, pred, scanCols, scanRow, inScan, forbStart, heap, pool, hypStride, total;
    // inside one hypothesis
    long long hu, hv, hc4r, hr4c, hforb;
    __host__ __device__ ExLayout(long long D, long long H)
    {
        auto up = [](long long x) { return (x + 63) & ~63ll; };
        // (up to 64 rows -- kbest_exact64_kernel -- the two index arrays are bytes: 1 216 instead of 1 600 bytes per hypothesis of a
        //  64 x 64 problem, so that the pools of 1 024 such problems at k = 200 fit the work space's budget side by side)
        const long long idx = D <= 64 ? 1 : 4;
        hu = 0; hv = up(8 * D); hc4r = hv + up(8 * D); hr4c = hc4r + up(idx * D); hforb = hr4c + up(idx * D);
        hypStride = up(hforb + D);
        long long o = 0;
        C = o;         o += up(8 * D * D);
        spc = o;       o += up(8 * D);
        pred = o;      o += up(4 * D);
        scanCols = o;  o += up(4 * D);
        scanRow = o;   o += up(D);
        inScan = o;    o += up(D);
        forbStart = o; o += up(D);
        heap = o;      o += up(16 * H);  // (gain, hypothesis | activeCol << 32) pairs: one load per comparison
        pool = o;      o += hypStride * H;
        total = up(o);
    }
};

struct Hyp {
    double *u, *v;
    int *c4r, *r4c;
    unsigned char *forb;
};

}  // namespace

long long exact_slot_bytes(int maxRow, int hypPerSlot) { return ExLayout(maxRow, hypPerSlot).total; }

// More than EXACT_LDS_ROWS (1 024) rows: one wave per problem, everything in the work space -- the lanes read each other's global
// stores, so every hand-over between them is a real fence (the heap is lane 0's alone; what the wave needs of it is broadcast from
// lane 0's registers).  The slowest form, for the sizes nothing else takes: the reference has no size limit (cpp:571-644).
__global__ void __launch_bounds__(64) kbest_exact_kernel(ExactParams p)
{
    const int lane = threadIdx.x;
    const double INF = d_inf();
    const ExLayout L(p.maxRow, p.hypPerSlot);
    unsigned char *ws = p.work + (long long)blockIdx.x * L.total;
    // the scratch of a search (ScratchSpace, hpp:73-142)
    double *spc = reinterpret_cast<double *>(ws + L.spc);
    int *pred = reinterpret_cast<int *>(ws + L.pred);
    int *scanCols = reinterpret_cast<int *>(ws + L.scanCols);
    unsigned char *scanRow = ws + L.scanRow, *inScan = ws + L.inScan, *forbStart = ws + L.forbStart;
    double *Cw = reinterpret_cast<double *>(ws + L.C);
    struct HeapE { double g; long long idx; };  // idx: hypothesis | activeCol << 32
    HeapE *heap = reinterpret_cast<HeapE *>(ws + L.heap);
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const bool maximize = p.maximize != 0;

    auto hyp = [&](int i) {
        unsigned char *b = ws + L.pool + (long long)i * L.hypStride;
        return Hyp{reinterpret_cast<double *>(b + L.hu), reinterpret_cast<double *>(b + L.hv), reinterpret_cast<int *>(b + L.hc4r),
                   reinterpret_cast<int *>(b + L.hr4c), b + L.hforb};
    };
    // a real fence: what one lane wrote to the work space, every lane reads after this
    auto full_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads(); };
    auto sync = [&]() { full_sync(); };
    auto first_i32 = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto first_f64 = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };

    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        const int N = p.nRow ? p.nRow[b] : p.maxRow, M = p.nCol ? p.nCol[b] : p.maxCol;
        const long long outBase = (long long)b * p.k;
        if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol) {  // undefined in the reference
            if (lane == 0) p.nf[b] = (M == 0 || N == 0) ? 0 : -1;
            continue;
        }
        const int D = N;
        const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.maxRow * p.maxCol);
        // ---- makeCostMatrixSafe (cpp:534-569) + zero padding (cpp:582-585, 663-666) ----
        double d = INF;
        for (long long i = lane; i < (long long)N * M; i += 64) {
            const double x = maximize ? -Cg[i] : Cg[i];
            d = min_keep(d, x);
        }
        d = wave_min_f64(d);  // min C, or min(-C) = -max C
        // (the reference: C - min C, or -C + max C; with d = -max C the second is (-C) - d)
        for (long long i = lane; i < (long long)N * M; i += 64) Cw[i] = (maximize ? -Cg[i] : Cg[i]) - d;
        for (long long i = (long long)N * M + lane; i < (long long)N * N; i += 64) Cw[i] = 0.0;
        const double CDelta = (maximize ? -d : d) * (double)M;  // cpp:583 (maximize: CDelta is max C)
        full_sync();
        // The pool: one record per hypothesis the reference's queue ever holds, handed out in order and never reused -- at most
        // 1 + (k - 1) M pushes (kbest_capi.cpp: plan_exact), so the records of a problem's pool are enough by construction.
        int nextHyp = 0, heapN = 0;  // (wave-uniform registers)

        // one shortest augmenting path from column `start`, the dual update, the path flip (cpp:146-230, 283-358, 82-117).
        // useForb: rows flagged in forbStart are skipped while the start column itself is scanned (cpp:310).  1 = infeasible.
        auto augment = [&](const Hyp &h, int start, bool useForb) -> int {
            for (int r = lane; r < D; r += 64) { scanRow[r] = 0; spc[r] = INF; }
            sync();
            int nScanned = 0, sink = -1, cur = start;
            double delta = 0.0;
            do {
                if (lane == 0) scanCols[nScanned] = cur;
                nScanned++;
                const double uc = h.u[cur];
                const double *Ccol = Cw + (long long)cur * D;
                const bool padded = cur >= M;  // (a zero-padded column, cpp:582-585: its costs are 0.0 -- no trip to memory for them)
                const bool forbNow = useForb && cur == start;
                double best = INF;
                int bestR = 0x7fffffff;
                for (int r = lane; r < D; r += 64) {
                    if (!inScan[r]) continue;
                    if (forbNow && forbStart[r]) continue;
                    const double rc = ((delta + (padded ? 0.0 : Ccol[r])) - uc) - h.v[r];  // cpp:183 / 313: left to right
                    double s = spc[r];
                    if (rc < s) { pred[r] = cur; spc[r] = rc; s = rc; }
                    if (s < best) { best = s; bestR = r; }  // (ascending r within the lane: the first minimum is the lowest row)
                }
                const double minVal = wave_min_f64(best);
                if (!(minVal < INF)) return 1;  // cpp:197-203, 327-334
                const int closest = wave_min_i32(best == minVal ? bestR : 0x7fffffff);  // lowest row among equal minima
                if (lane == 0) { scanRow[closest] = 1; inScan[closest] = 0; }
                sync();
                delta = spc[closest];
                const int col = h.c4r[closest];
                if (col == -1) sink = closest; else cur = col;
            } while (sink == -1);
            // updateDualAndAugment (cpp:82-117): u of the start column, u of the other scanned columns, v of the scanned rows
            for (int i = lane; i < nScanned; i += 64) {
                const int c = scanCols[i];
                if (i == 0) h.u[c] = h.u[c] + delta;
                else h.u[c] = h.u[c] + delta - spc[h.r4c[c]];
            }
            for (int r = lane; r < D; r += 64)
                if (scanRow[r]) h.v[r] = h.v[r] - delta + spc[r];
            sync();
            if (lane == 0) {
                int r = sink, c;
                do {
                    c = pred[r];
                    h.c4r[r] = c;
                    const int nxt = h.r4c[c];
                    h.r4c[c] = r;
                    r = nxt;
                } while (c != start);
            }
            sync();
            return 0;
        };
        // calcGain (cpp:59-80): serial, left to right, from 0.0 (every lane the same sum)
        // (the terms of 64 columns are fetched by the lanes at once; the additions stay one after the other, in column order)
        auto gain_of = [&](const Hyp &h, int nCol4Gain) {
            double g = 0.0;
            for (int c0 = 0; c0 < nCol4Gain; c0 += 64) {
                const int c = c0 + lane;
                const double term = c < nCol4Gain ? Cw[(long long)c * D + h.r4c[c]] : 0.0;
                const int n = nCol4Gain - c0 < 64 ? nCol4Gain - c0 : 64;
                for (int j = 0; j < n; j++) g = g + readlane_f64(term, j);
            }
            return g;
        };
        // std::priority_queue<pMurtyHyp> with a < b <=> a.gain > b.gain (cpp:35-37): libstdc++'s __push_heap / __adjust_heap.
        // Lane 0's alone: the heap's size is kept by every lane (heapN), its entries are read and written by lane 0 only.
        auto sift_up = [&](int hole, int top, HeapE val) {  // (lane 0)
            int parent = (hole - 1) / 2;
            while (hole > top) {
                const HeapE pe = heap[parent];
                if (!(pe.g > val.g)) break;
                heap[hole] = pe;
                hole = parent;
                parent = (hole - 1) / 2;
            }
            heap[hole] = val;
        };
        auto heap_push = [&](int hidx, int act, double g) {
            if (lane == 0) sift_up(heapN, 0, HeapE{g, (long long)hidx | ((long long)act << 32)});
            heapN++;
        };
        // the top entry, in every lane's registers
        auto heap_top = [&](double &g, int &hidx, int &act) {
            HeapE e{0.0, 0};
            if (lane == 0) e = heap[0];
            g = first_f64(e.g);
            hidx = first_i32((int)(e.idx & 0xffffffffll));
            act = first_i32((int)(e.idx >> 32));
        };
        auto heap_pop = [&]() {  // (the caller has taken the top with heap_top)
            const int len = heapN - 1;
            if (lane == 0 && len > 0) {
                const HeapE val = heap[len];
                int hole = 0, child = 0;
                while (child < (len - 1) / 2) {
                    child = 2 * (child + 1);
                    HeapE ce = heap[child];
                    const HeapE le = heap[child - 1];  // (two independent loads)
                    if (ce.g > le.g) { child--; ce = le; }
                    heap[hole] = ce;
                    hole = child;
                }
                if ((len & 1) == 0 && child == (len - 2) / 2) {
                    child = 2 * (child + 1);
                    heap[hole] = heap[child - 1];
                    hole = child - 1;
                }
                sift_up(hole, 0, val);
            }
            heapN = len;
        };
        auto emit = [&](int hidx, double g, int slot) {
            const Hyp h = hyp(hidx);
            for (int c = lane; c < M; c += 64) put_index(p.row4col, (outBase + slot) * p.ldCol + c, h.r4c[c], tabI8);
            if (p.col4row)
                for (int r = lane; r < N; r += 64) put_index(p.col4row, (outBase + slot) * p.ldRow + r, h.c4r[r], tabI8);
            const double out = maximize ? (-g + CDelta) : (g + CDelta);  // cpp:626-630
            if (lane == 0) p.gain[outBase + slot] = out;
            return out;
        };

        // ---- root: shortestPathCPP (cpp:119-238), N augmentations in column order on the padded problem ----
        if (p.hypPerSlot < 2) { if (lane == 0) p.nf[b] = -4; continue; }
        const Hyp hrG = hyp(nextHyp);
        const Hyp hr = hrG;
        for (int i = lane; i < D; i += 64) { hr.c4r[i] = -1; hr.r4c[i] = -1; hr.u[i] = 0.0; hr.v[i] = 0.0; hr.forb[i] = 0; }
        sync();
        int infeasible = 0;
        for (int c = 0; c < D && !infeasible; c++) {
            for (int r = lane; r < D; r += 64) inScan[r] = 1;
            sync();
            infeasible = augment(hr, c, false);
        }
        if (infeasible) {  // kBest2D returns 0 (cpp:588-593)
            if (lane == 0) p.nf[b] = 0;
            continue;
        }
        const double rootGain = gain_of(hr, M);
        if (lane == 0) hr.forb[hr.r4c[0]] = 1;  // cpp:232-235
        sync();
        const double gain0 = emit(nextHyp, rootGain, 0);
        const double cutoffGain = maximize ? (rootGain - p.cutoff) : (rootGain + p.cutoff);  // cpp:680-686
        heap_push(nextHyp, 0, rootGain);
        nextHyp++;
        long long pushed = 0;
        int sweep = 1, err = 0;
        for (; sweep < p.k; sweep++) {  // cpp:607-634
            double gTop;
            int cur, a;
            heap_top(gTop, cur, a);
            heap_pop();
            // the hypothesis that is split: read where it lies, behind a real fence
            const Hyp hpG = hyp(cur);
            full_sync();
            const Hyp hp = hpG;
            // ---- split (cpp:455-532): the children of columns a .. M-1, each fully solved, pushed in that order ----
            for (int c = a; c < M; c++) {
                // rows still owned by columns >= c of the parent (cpp:480-488; 506-508, 512, 525-527)
                for (int r = lane; r < D; r += 64) { inScan[r] = 0; forbStart[r] = (c == a) ? hp.forb[r] : 0; }  // cpp:490 / 510
                sync();
                for (int j = c + lane; j < D; j += 64) inScan[hp.r4c[j]] = 1;
                if (c != a && lane == 0) forbStart[hp.r4c[c]] = 1;  // cpp:516
                sync();
                if (nextHyp >= p.hypPerSlot) { err = 1; break; }  // (cannot happen: see the pool above)
                const Hyp hcG = hyp(nextHyp);
                const Hyp hc = hcG;
                for (int i = lane; i < D; i += 64) {  // shortestPathUpdateCPP, cpp:262-278
                    hc.r4c[i] = hp.r4c[i]; hc.c4r[i] = hp.c4r[i]; hc.u[i] = hp.u[i]; hc.v[i] = hp.v[i]; hc.forb[i] = forbStart[i];
                }
                sync();
                if (lane == 0) { hc.c4r[hc.r4c[c]] = -1; hc.r4c[c] = -1; }
                sync();
                if (augment(hc, c, true)) continue;  // infeasible child: gain -1, dropped (cpp:496, 521); its record is the next child's
                const double g = gain_of(hc, M);
                const bool cut = p.useCutoff && (maximize ? (g < cutoffGain) : (g > cutoffGain));  // cutHyp, hpp:130-131
                if (cut) continue;
                if (lane == 0) hc.forb[hc.r4c[c]] = 1;  // cpp:362
                sync();
                heap_push(nextHyp, c, g);
                nextHyp++;
                pushed++;
            }
            if (err || heapN == 0) break;
            heap_top(gTop, cur, a);
            const double gs = emit(cur, gTop, sweep);
            if (p.useCutoff) {  // cpp:709-719
                if (!maximize) { if (gs > gain0 + p.cutoff) break; }
                else           { if (gs < gain0 - p.cutoff) break; }
            }
        }
        if (lane == 0) {
            p.nf[b] = err ? -4 : sweep;
            if (p.pushed) p.pushed[b] = pushed;
        }
        full_sync();  // (the next problem of this slot reuses the work space)
    }
}

// ---- 65 .. 1 024 rows, NW waves per problem ------------------------------------------------------------------------------------
// kbest_exact_kernel<1>'s sequence of operations with a sweep's children dealt to the waves, as kbest_exact64_kernel below does it:
// every child-solving wave has its own scratch of a search and its own hypothesis-being-solved in LDS, the hypothesis being split is
// loaded once per sweep by all of them, wave 0 makes the pushes in column order behind a barrier; the ROOT's N augmentations -- most of
// the run on problems of many rows and few columns -- are shared by all eight waves, a Dijkstra step's rows dealt to the threads.
// (One problem of 200 x 150, k = 50: 446 ms on one wave, 71 ms here; two of 1 000 x 12, k = 10: 1.80 s / 0.76 s.)
template <int NW>
__global__ void __launch_bounds__(64 * NW) kbest_exactN_kernel(ExactParams p)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = 64 * NW;
    const double INF = d_inf();
    const ExLayout L(p.maxRow, p.hypPerSlot);
    unsigned char *ws = p.work + (long long)blockIdx.x * L.total;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const long long Dm = p.maxRow;
    const long long scrB = (19 * Dm + 63) & ~63ll, hypB = (25 * Dm + 63) & ~63ll;
    // CW of the NW waves solve children (as many as the LDS holds scratch + hypothesis for); ALL of them share the rows of the root's
    // Dijkstra steps (augment_all below: on wave 0's scratch and hypothesis)
    const int CW = p.childWaves;
    unsigned char *my = lds + (wave < CW ? wave : 0) * (scrB + hypB);
    // this wave's scratch of a search (ScratchSpace, hpp:73-142) and the hypothesis it is solving
    double *spc = reinterpret_cast<double *>(my);
    int *pred = reinterpret_cast<int *>(my + 8 * Dm);
    int *scanCols = reinterpret_cast<int *>(my + 12 * Dm);
    unsigned char *scanRow = my + 16 * Dm, *inScan = my + 17 * Dm, *forbStart = my + 18 * Dm;
    auto lds_hyp = [&](unsigned char *o) {
        return Hyp{reinterpret_cast<double *>(o), reinterpret_cast<double *>(o + 8 * Dm), reinterpret_cast<int *>(o + 16 * Dm),
                   reinterpret_cast<int *>(o + 20 * Dm), o + 24 * Dm};
    };
    const Hyp hLds = lds_hyp(my + scrB);
    // everybody's: the hypothesis being split, the sweep's results by column, wave 0's words
    unsigned char *sh = lds + CW * (scrB + hypB);
    const Hyp pLds = lds_hyp(sh);
    double *resG = reinterpret_cast<double *>(sh + hypB);
    int *resIdx = reinterpret_cast<int *>(resG + ((p.maxCol + 1) & ~1));
    int *bc = resIdx + ((p.maxCol + 3) & ~3);
    double *red = reinterpret_cast<double *>(bc + 16);  // [2][NW] minima of the waves (double-buffered by step)
    int *redR = reinterpret_cast<int *>(red + 2 * NW);   // [2][NW] ... and their rows
    // wave 0's scratch and hypothesis, as everybody sees them (the root)
    double *spc0 = reinterpret_cast<double *>(lds);
    int *pred0 = reinterpret_cast<int *>(lds + 8 * Dm), *scanCols0 = reinterpret_cast<int *>(lds + 12 * Dm);
    unsigned char *scanRow0 = lds + 16 * Dm, *inScan0 = lds + 17 * Dm;
    enum { BC_CUR = 0, BC_ACT = 1, BC_STOP = 2, BC_NEXT = 3, BC_ERR = 4 };
    double *Cw = reinterpret_cast<double *>(ws + L.C);
    struct HeapE { double g; long long idx; };  // idx: hypothesis | activeCol << 32
    HeapE *heap = reinterpret_cast<HeapE *>(ws + L.heap);
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const bool maximize = p.maximize != 0;
    auto hyp = [&](int i) {
        unsigned char *b = ws + L.pool + (long long)i * L.hypStride;
        return Hyp{reinterpret_cast<double *>(b + L.hu), reinterpret_cast<double *>(b + L.hv), reinterpret_cast<int *>(b + L.hc4r),
                   reinterpret_cast<int *>(b + L.hr4c), b + L.hforb};
    };
    auto sync = [&]() { wave_fence(); };  // (hand-overs between the lanes of a wave through its own LDS)
    auto first_i32 = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto first_f64 = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };

    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        const int N = p.nRow ? p.nRow[b] : p.maxRow, M = p.nCol ? p.nCol[b] : p.maxCol;
        const long long outBase = (long long)b * p.k;
        if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol) {  // undefined in the reference
            if (threadIdx.x == 0) p.nf[b] = (M == 0 || N == 0) ? 0 : -1;
            continue;
        }
        const int D = N;
        const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.maxRow * p.maxCol);
        // ---- makeCostMatrixSafe (cpp:534-569) + zero padding (cpp:582-585, 663-666) ----
        double d = INF;
        for (long long i = threadIdx.x; i < (long long)N * M; i += NT) d = min_keep(d, maximize ? -Cg[i] : Cg[i]);
        d = wave_min_f64(d);
        if (lane == 0) red[wave] = d;
        __syncthreads();
        d = red[0];
        for (int w = 1; w < NW; w++) d = min_keep(d, red[w]);
        for (long long i = threadIdx.x; i < (long long)N * M; i += NT) Cw[i] = (maximize ? -Cg[i] : Cg[i]) - d;
        for (long long i = (long long)N * M + threadIdx.x; i < (long long)N * N; i += NT) Cw[i] = 0.0;
        const double CDelta = (maximize ? -d : d) * (double)M;  // cpp:583 (maximize: CDelta is max C)
        __syncthreads();
        int heapN = 0;  // (wave 0's)

        // one shortest augmenting path from column `start`, the dual update, the path flip (cpp:146-230, 283-358, 82-117).
        // useForb: rows flagged in forbStart are skipped while the start column itself is scanned (cpp:310).  1 = infeasible.
        auto augment = [&](const Hyp &h, int start, bool useForb) -> int {
            for (int r = lane; r < D; r += 64) { scanRow[r] = 0; spc[r] = INF; }
            sync();
            int nScanned = 0, sink = -1, cur = start;
            double delta = 0.0;
            do {
                if (lane == 0) scanCols[nScanned] = cur;
                nScanned++;
                const double uc = h.u[cur];
                const double *Ccol = Cw + (long long)cur * D;
                const bool padded = cur >= M;  // (a zero-padded column, cpp:582-585: its costs are 0.0 -- no trip to memory for them)
                const bool forbNow = useForb && cur == start;
                double best = INF;
                int bestR = 0x7fffffff;
                for (int r = lane; r < D; r += 64) {
                    if (!inScan[r]) continue;
                    if (forbNow && forbStart[r]) continue;
                    const double rc = ((delta + (padded ? 0.0 : Ccol[r])) - uc) - h.v[r];  // cpp:183 / 313: left to right
                    double s = spc[r];
                    if (rc < s) { pred[r] = cur; spc[r] = rc; s = rc; }
                    if (s < best) { best = s; bestR = r; }  // (ascending r within the lane: the first minimum is the lowest row)
                }
                const double minVal = wave_min_f64(best);
                if (!(minVal < INF)) return 1;  // cpp:197-203, 327-334
                const int closest = wave_min_i32(best == minVal ? bestR : 0x7fffffff);  // lowest row among equal minima
                if (lane == 0) { scanRow[closest] = 1; inScan[closest] = 0; }
                sync();
                delta = spc[closest];
                const int col = h.c4r[closest];
                if (col == -1) sink = closest; else cur = col;
            } while (sink == -1);
            // updateDualAndAugment (cpp:82-117)
            for (int i = lane; i < nScanned; i += 64) {
                const int c = scanCols[i];
                if (i == 0) h.u[c] = h.u[c] + delta;
                else h.u[c] = h.u[c] + delta - spc[h.r4c[c]];
            }
            for (int r = lane; r < D; r += 64)
                if (scanRow[r]) h.v[r] = h.v[r] - delta + spc[r];
            sync();
            if (lane == 0) {
                int r = sink, c;
                do {
                    c = pred[r];
                    h.c4r[r] = c;
                    const int nxt = h.r4c[c];
                    h.r4c[c] = r;
                    r = nxt;
                } while (c != start);
            }
            sync();
            return 0;
        };
        // calcGain (cpp:59-80): serial, left to right, from 0.0 (the terms of 64 columns fetched by the lanes at once)
        auto gain_of = [&](const Hyp &h, int nCol4Gain) {
            double g = 0.0;
            for (int c0 = 0; c0 < nCol4Gain; c0 += 64) {
                const int c = c0 + lane;
                const double term = c < nCol4Gain ? Cw[(long long)c * D + h.r4c[c]] : 0.0;
                const int n = nCol4Gain - c0 < 64 ? nCol4Gain - c0 : 64;
                for (int j = 0; j < n; j++) g = g + readlane_f64(term, j);
            }
            return g;
        };
        // libstdc++'s __push_heap / __adjust_heap (wave 0, lane 0)
        auto sift_up = [&](int hole, HeapE val) {
            int parent = (hole - 1) / 2;
            while (hole > 0) {
                const HeapE pe = heap[parent];
                if (!(pe.g > val.g)) break;
                heap[hole] = pe;
                hole = parent;
                parent = (hole - 1) / 2;
            }
            heap[hole] = val;
        };
        auto heap_push = [&](int hidx, int act, double g) {
            if (lane == 0) sift_up(heapN, HeapE{g, (long long)hidx | ((long long)act << 32)});
            heapN++;
        };
        auto heap_top = [&](double &g, int &hidx, int &act) {
            HeapE e{0.0, 0};
            if (lane == 0) e = heap[0];
            g = first_f64(e.g);
            hidx = first_i32((int)(e.idx & 0xffffffffll));
            act = first_i32((int)(e.idx >> 32));
        };
        auto heap_pop = [&]() {
            const int len = heapN - 1;
            if (lane == 0 && len > 0) {
                const HeapE val = heap[len];
                int hole = 0, child = 0;
                while (child < (len - 1) / 2) {
                    child = 2 * (child + 1);
                    HeapE ce = heap[child];
                    const HeapE le = heap[child - 1];
                    if (ce.g > le.g) { child--; ce = le; }
                    heap[hole] = ce;
                    hole = child;
                }
                if ((len & 1) == 0 && child == (len - 2) / 2) {
                    child = 2 * (child + 1);
                    heap[hole] = heap[child - 1];
                    hole = child - 1;
                }
                sift_up(hole, val);
            }
            heapN = len;
        };
        // a hypothesis between a wave's LDS and its record in the pool (other waves read a record only behind a barrier)
        auto copy_hyp = [&](const Hyp &src, const Hyp &dst) {
            for (int i = lane; i < D; i += 64) { dst.r4c[i] = src.r4c[i]; dst.c4r[i] = src.c4r[i]; dst.u[i] = src.u[i]; dst.v[i] = src.v[i]; dst.forb[i] = src.forb[i]; }
        };
        auto emit = [&](int hidx, double g, int slot) {
            const Hyp h = hyp(hidx);
            for (int c = lane; c < M; c += 64) put_index(p.row4col, (outBase + slot) * p.ldCol + c, h.r4c[c], tabI8);
            if (p.col4row)
                for (int r = lane; r < N; r += 64) put_index(p.col4row, (outBase + slot) * p.ldRow + r, h.c4r[r], tabI8);
            const double out = maximize ? (-g + CDelta) : (g + CDelta);  // cpp:626-630
            if (lane == 0) p.gain[outBase + slot] = out;
            return out;
        };
        double gain0 = 0.0;
        long long pushed = 0;
        // wave 0, behind a sweep's children (and behind the root): the pushes in column order, the new top out, the next hypothesis
        // popped -- or the end of the problem (BC_STOP: the number of solutions + 1; -1: the pool ran out)
        auto turn = [&](int sweep, int a, bool root) {
            if (!root) {
                for (int c = a; c < M; c++) {
                    const int idx = resIdx[c];
                    if (idx < 0) continue;
                    heap_push(idx, c, resG[c]);
                    pushed++;
                }
            }
            int stop = 0;
            double gTop;
            int cur, act;
            if (bc[BC_ERR]) stop = -1;
            else if (heapN == 0) stop = sweep + 1;
            else if (!root) {
                heap_top(gTop, cur, act);
                const double gs = emit(cur, gTop, sweep);
                if (p.useCutoff && (maximize ? (gs < gain0 - p.cutoff) : (gs > gain0 + p.cutoff))) stop = sweep + 1;  // cpp:709-719
                else if (sweep + 1 >= p.k) stop = sweep + 2;
            } else if (p.k <= 1) stop = 2;
            if (!stop) {
                heap_top(gTop, cur, act);
                heap_pop();
                if (lane == 0) { bc[BC_CUR] = cur; bc[BC_ACT] = act; }
            }
            if (lane == 0) bc[BC_STOP] = stop;
        };

        // ---- root: shortestPathCPP (cpp:119-238), N augmentations in column order on the padded problem.  EVERY wave takes part:
        // the rows of a Dijkstra step are shared by all threads (row = thread, thread + 64 NW, ...), each wave reduces its rows, the
        // waves' minima meet in LDS (the lowest row among equal minima, as everywhere) -- two barriers per step.  On problems of
        // hundreds of rows the root's N augmentations on ONE wave were most of the run (2 x 1 000x12, k = 10: 1.6 of 1.7 s).
        auto augment_all = [&](const Hyp &h, int start) -> int {
            for (int r = threadIdx.x; r < D; r += NT) { scanRow0[r] = 0; spc0[r] = INF; }
            __syncthreads();
            int nScanned = 0, sink = -1, cur = start, par = 0;
            double delta = 0.0;
            do {
                if (threadIdx.x == 0) scanCols0[nScanned] = cur;
                nScanned++;
                const double uc = h.u[cur];
                const double *Ccol = Cw + (long long)cur * D;
                const bool padded = cur >= M;  // (a zero-padded column, cpp:582-585: its costs are 0.0 -- no trip to memory for them)
                double best = INF;
                int bestR = 0x7fffffff;
                for (int r = threadIdx.x; r < D; r += NT) {
                    if (!inScan0[r]) continue;
                    const double rc = ((delta + (padded ? 0.0 : Ccol[r])) - uc) - h.v[r];  // cpp:183: left to right
                    double sv = spc0[r];
                    if (rc < sv) { pred0[r] = cur; spc0[r] = rc; sv = rc; }
                    if (sv < best) { best = sv; bestR = r; }  // (ascending r within the thread: the first minimum is its lowest row)
                }
                const double wmin = wave_min_f64(best);
                const int wrow = wave_min_i32(best == wmin ? bestR : 0x7fffffff);
                if (lane == 0) { red[par * NW + wave] = wmin; redR[par * NW + wave] = wrow; }
                __syncthreads();
                double minVal = red[par * NW];
                int closest = redR[par * NW];
                for (int w = 1; w < NW; w++) {
                    const double vw = red[par * NW + w];
                    const int rw = redR[par * NW + w];
                    if (vw < minVal || (vw == minVal && rw < closest)) { minVal = vw; closest = rw; }
                }
                par ^= 1;
                if (!(minVal < INF)) return 1;  // cpp:197-203 (the same value in every thread)
                if (threadIdx.x == 0) { scanRow0[closest] = 1; inScan0[closest] = 0; }
                __syncthreads();
                delta = spc0[closest];
                const int col = h.c4r[closest];
                if (col == -1) sink = closest; else cur = col;
            } while (sink == -1);
            // updateDualAndAugment (cpp:82-117)
            for (int i = threadIdx.x; i < nScanned; i += NT) {
                const int c = scanCols0[i];
                if (i == 0) h.u[c] = h.u[c] + delta;
                else h.u[c] = h.u[c] + delta - spc0[h.r4c[c]];
            }
            for (int r = threadIdx.x; r < D; r += NT)
                if (scanRow0[r]) h.v[r] = h.v[r] - delta + spc0[r];
            __syncthreads();
            if (threadIdx.x == 0) {
                int r = sink, c;
                do {
                    c = pred0[r];
                    h.c4r[r] = c;
                    const int nxt = h.r4c[c];
                    h.r4c[c] = r;
                    r = nxt;
                } while (c != start);
            }
            __syncthreads();
            return 0;
        };
        {
            const Hyp hr = lds_hyp(lds + scrB);  // (wave 0's hypothesis-being-solved)
            if (threadIdx.x == 0) { bc[BC_ERR] = 0; bc[BC_NEXT] = 1; }
            for (int i = threadIdx.x; i < D; i += NT) { hr.c4r[i] = -1; hr.r4c[i] = -1; hr.u[i] = 0.0; hr.v[i] = 0.0; hr.forb[i] = 0; }
            __syncthreads();
            int infeasible = 0;
            for (int c = 0; c < D && !infeasible; c++) {
                for (int r = threadIdx.x; r < D; r += NT) inScan0[r] = 1;
                __syncthreads();
                infeasible = augment_all(hr, c);
            }
            if (wave == 0) {
                if (infeasible) {  // kBest2D returns 0 (cpp:588-593)
                    if (lane == 0) bc[BC_STOP] = 1;
                } else if (p.hypPerSlot < 2) {
                    if (lane == 0) bc[BC_STOP] = -1;
                } else {
                    const double rootGain = gain_of(hr, M);
                    if (lane == 0) hr.forb[hr.r4c[0]] = 1;  // cpp:232-235
                    sync();
                    copy_hyp(hr, hyp(0));
                    gain0 = emit(0, rootGain, 0);
                    if (lane == 0) resG[0] = rootGain;
                    heap_push(0, 0, rootGain);
                    turn(0, 0, true);
                }
            }
        }
        __syncthreads();
        const double cutoffGain = maximize ? (resG[0] - p.cutoff) : (resG[0] + p.cutoff);  // cpp:680-686 (every wave's copy)
        int stop = bc[BC_STOP];
        __syncthreads();  // (resG[0] is a child's slot from here on)
        for (int sweep = 1; !stop; sweep++) {  // cpp:607-634
            const int cur = bc[BC_CUR], a = bc[BC_ACT];
            // the hypothesis that is split: from the pool into LDS, by everybody, once
            {
                const Hyp hpG = hyp(cur);
                for (int i = threadIdx.x; i < D; i += NT) { pLds.r4c[i] = hpG.r4c[i]; pLds.c4r[i] = hpG.c4r[i]; pLds.u[i] = hpG.u[i]; pLds.v[i] = hpG.v[i]; pLds.forb[i] = hpG.forb[i]; }
            }
            __syncthreads();
            const Hyp hp = pLds, hc = hLds;
            // ---- split (cpp:455-532): the children of columns a .. M-1, each fully solved; this wave's share ----
            for (int c = wave < CW ? a + wave : M; c < M; c += CW) {
                // rows still owned by columns >= c of the parent (cpp:480-488; 506-508, 512, 525-527)
                for (int r = lane; r < D; r += 64) { inScan[r] = 0; forbStart[r] = (c == a) ? hp.forb[r] : 0; }  // cpp:490 / 510
                sync();
                for (int j = c + lane; j < D; j += 64) inScan[hp.r4c[j]] = 1;
                if (c != a && lane == 0) forbStart[hp.r4c[c]] = 1;  // cpp:516
                sync();
                for (int i = lane; i < D; i += 64) {  // shortestPathUpdateCPP, cpp:262-278
                    hc.r4c[i] = hp.r4c[i]; hc.c4r[i] = hp.c4r[i]; hc.u[i] = hp.u[i]; hc.v[i] = hp.v[i]; hc.forb[i] = forbStart[i];
                }
                sync();
                if (lane == 0) { hc.c4r[hc.r4c[c]] = -1; hc.r4c[c] = -1; }
                sync();
                int idx = -1;
                if (!augment(hc, c, true)) {  // else infeasible: gain -1, dropped (cpp:496, 521)
                    const double g = gain_of(hc, M);
                    if (!(p.useCutoff && (maximize ? (g < cutoffGain) : (g > cutoffGain)))) {  // cutHyp, hpp:130-131
                        if (lane == 0) hc.forb[hc.r4c[c]] = 1;  // cpp:362
                        sync();
                        if (lane == 0) idx = atomicAdd(&bc[BC_NEXT], 1);
                        idx = first_i32(idx);
                        if (idx >= p.hypPerSlot) {  // (cannot happen: at most 1 + (k - 1) M pushes, plan_exact)
                            if (lane == 0) bc[BC_ERR] = 1;
                            idx = -1;
                        } else {
                            copy_hyp(hc, hyp(idx));
                            if (lane == 0) resG[c] = g;
                        }
                    }
                }
                if (lane == 0) resIdx[c] = idx;
            }
            __syncthreads();
            if (wave == 0) turn(sweep, a, false);
            __syncthreads();
            stop = bc[BC_STOP];
        }
        if (threadIdx.x == 0) {
            p.nf[b] = stop < 0 ? -4 : stop - 1;
            if (p.pushed) p.pushed[b] = pushed;
        }
        __syncthreads();
    }
}

// ---- up to 64 rows: the same algorithm with the hypothesis in REGISTERS --------------------------------------------------------
// lane = row for v / col4row, lane = column for u / row4col, the forbidden rows one 64-bit mask, the padded cost copy and u in LDS:
// the Dijkstra step is then the hand-written loop of the LDS kernels (kbest_lap.h: bit-exact against the reference's duals, tests/
// test_assign_golden.py), the dual update and the path flip are lane reads, and a child is a handful of register copies of its
// parent.  What the general kernel above spends per step (a pass over LDS arrays, two reductions, three hand-overs: ~1 000 cycles of
// one wave) is ~150 here.  Everything the ORDER depends on is unchanged: the N augmentations of the padded root in column order,
// the children of columns activeCol .. M-1 solved completely and pushed in that order, libstdc++'s heap.
// Lane 0 keeps the heap in HBM; the entry a push will be compared with first -- the parent of the next free position -- is
// fetched right after the push before, so the usual push (no move, or one) does not wait for memory.
// HEAP_LDS: the heap too lies in LDS -- where 16 bytes per hypothesis of the pool fit beside the cost copy in a quarter of a CU's LDS
// (a 28 x 10 frame at k = 200: 2 000 entries, 32 KB): a pop is then eleven LDS round trips instead of eleven trips to HBM.
// NW waves per problem: the children of a sweep -- each solved completely from the parent, independent of one another -- are dealt to
// the waves (child of column a + w, a + w + NW, ...); only the PUSHES have an order, and wave 0 makes them in column order once all
// children are solved (their gains wait in LDS), then emits the new top and pops the next hypothesis: two barriers per sweep.  The
// pool's records are handed out by an LDS counter: which record a hypothesis gets does not matter, only its place in the heap does.
template <bool HEAP_LDS, int NW>
__global__ void __launch_bounds__(64 * NW) kbest_exact64_kernel(ExactParams p)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const double INF = d_inf();
    const ExLayout L(p.maxRow, p.hypPerSlot);
    unsigned char *ws = p.work + (long long)blockIdx.x * L.total;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double *uL = reinterpret_cast<double *>(lds) + wave * 136;  // duals of the columns of the hypothesis this wave is solving
    double *gainW = uL + 64;                                     // serial_gain's line of terms (this wave's)
    double *resG = reinterpret_cast<double *>(lds) + NW * 136;   // [64] gains of the sweep's children, by column
    int *resIdx = reinterpret_cast<int *>(resG + 64);            // [64] their records, -1 = dropped
    int *bc = resIdx + 64;                                       // [16] wave 0's words for everybody: see below
    double *red = reinterpret_cast<double *>(bc + 16);           // [NW] cross-wave minimum of the set-up
    double *Cs = red + ((NW + 1) & ~1);                          // padded, shifted costs: Cs[c * LDC + r]
    struct HeapE { double g; long long idx; };                   // idx: hypothesis | activeCol << 32
    HeapE *heap = HEAP_LDS ? reinterpret_cast<HeapE *>(Cs + (((long long)p.maxRow * (p.maxRow | 1) + 1) & ~1ll)) : reinterpret_cast<HeapE *>(ws + L.heap);
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const bool maximize = p.maximize != 0;
    enum { BC_CUR = 0, BC_ACT = 1, BC_STOP = 2, BC_NEXT = 3, BC_ERR = 4 };
    struct Hyp8 {  // (the record of a problem of up to 64 rows: byte indices)
        double *u, *v;
        signed char *c4r, *r4c;
        unsigned char *forb;
    };
    auto hyp = [&](int i) {
        unsigned char *b = ws + L.pool + (long long)i * L.hypStride;
        return Hyp8{reinterpret_cast<double *>(b + L.hu), reinterpret_cast<double *>(b + L.hv), reinterpret_cast<signed char *>(b + L.hc4r),
                    reinterpret_cast<signed char *>(b + L.hr4c), b + L.hforb};
    };
    auto first_f64 = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };

    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        const int N = p.nRow ? p.nRow[b] : p.maxRow, M = p.nCol ? p.nCol[b] : p.maxCol;
        const long long outBase = (long long)b * p.k;
        if (N < 1 || M < 1 || N < M || N > p.maxRow || M > p.maxCol || N > 64) {  // undefined in the reference
            if (threadIdx.x == 0) p.nf[b] = (M == 0 || N == 0) ? 0 : -1;
            continue;
        }
        const int D = N, LDC = D | 1;
        const int rl = lane < D ? lane : D - 1;
        const u64 allRows = (D >= 64) ? ~0ull : ((1ull << D) - 1ull);
        const double *Cg = p.cost + (p.costOff ? p.costOff[b] : (long long)b * p.maxRow * p.maxCol);
        // ---- makeCostMatrixSafe (cpp:534-569) + zero padding (cpp:582-585, 663-666) ----
        double d = INF;
        for (int i = threadIdx.x; i < N * M; i += 64 * NW) d = min_keep(d, maximize ? -Cg[i] : Cg[i]);
        d = wave_min_f64(d);
        if (NW > 1) {
            if (lane == 0) red[wave] = d;
            __syncthreads();
            d = red[0];
            for (int w = 1; w < NW; w++) d = min_keep(d, red[w]);
        }
        // d = min C, or min(-C) = -max C
        for (int c = wave; c < D; c += NW)
            if (lane < D) Cs[c * LDC + lane] = c < M ? (maximize ? -Cg[c * N + lane] : Cg[c * N + lane]) - d : 0.0;
        const double CDelta = (maximize ? -d : d) * (double)M;  // cpp:583 (maximize: CDelta is max C)
        __syncthreads();
        int heapN = 0;             // (wave 0's; wave-uniform)
        HeapE nextParent{0.0, 0};  // wave 0, lane 0: heap[(heapN - 1) / 2], fetched ahead

        auto sift_up = [&](int hole, HeapE val, bool havePe, HeapE pe0) {  // (lane 0; libstdc++ __push_heap)
            int parent = (hole - 1) / 2;
            bool first = havePe;
            while (hole > 0) {
                const HeapE pe = first ? pe0 : heap[parent];
                first = false;
                if (!(pe.g > val.g)) break;
                heap[hole] = pe;
                hole = parent;
                parent = (hole - 1) / 2;
            }
            heap[hole] = val;
        };
        auto fetch_ahead = [&]() { if (!HEAP_LDS && lane == 0 && heapN > 0) nextParent = heap[(heapN - 1) / 2]; };
        auto heap_push = [&](int hidx, int act, double g) {
            if (lane == 0) sift_up(heapN, HeapE{g, (long long)hidx | ((long long)act << 32)}, !HEAP_LDS, nextParent);
            heapN++;
            fetch_ahead();
        };
        auto heap_top = [&](double &g, int &hidx, int &act) {
            HeapE e{0.0, 0};
            if (lane == 0) e = heap[0];
            g = first_f64(e.g);
            hidx = __builtin_amdgcn_readfirstlane((int)(e.idx & 0xffffffffll));
            act = __builtin_amdgcn_readfirstlane((int)(e.idx >> 32));
        };
        auto heap_pop = [&]() {  // (libstdc++ __adjust_heap; the caller has taken the top with heap_top)
            const int len = heapN - 1;
            if (lane == 0 && len > 0) {
                const HeapE val = heap[len];
                int hole = 0, child = 0;
                while (child < (len - 1) / 2) {
                    child = 2 * (child + 1);
                    HeapE ce = heap[child];
                    const HeapE le = heap[child - 1];  // (two independent loads)
                    if (ce.g > le.g) { child--; ce = le; }
                    heap[hole] = ce;
                    hole = child;
                }
                if ((len & 1) == 0 && child == (len - 2) / 2) {
                    child = 2 * (child + 1);
                    heap[hole] = heap[child - 1];
                    hole = child - 1;
                }
                sift_up(hole, val, false, val);
            }
            heapN = len;
            fetch_ahead();
        };
        // a hypothesis and its record in the pool: element i of every array on lane i, both ways (whichever wave: a record is read
        // by other waves only behind a barrier)
        auto store_hyp = [&](int hidx, double v, int c4r, int r4c, u64 forb) {
            const Hyp8 h = hyp(hidx);
            if (lane < D) { h.u[lane] = uL[lane]; h.v[lane] = v; h.c4r[lane] = (signed char)c4r; h.r4c[lane] = (signed char)r4c; h.forb[lane] = (unsigned char)((forb >> lane) & 1ull); }
        };
        auto emit = [&](int hidx, double g, int slot) {
            const Hyp8 h = hyp(hidx);
            if (lane < M) put_index(p.row4col, (outBase + slot) * p.ldCol + lane, h.r4c[lane], tabI8);
            if (p.col4row && lane < N) put_index(p.col4row, (outBase + slot) * p.ldRow + lane, h.c4r[lane], tabI8);
            const double out = maximize ? (-g + CDelta) : (g + CDelta);  // cpp:626-630
            if (lane == 0) p.gain[outBase + slot] = out;
            return out;
        };

        double v = 0.0, spc, delta;
        int c4r = -1, r4c = -1, pred, sink = 0;
        u64 scanned;
        double gain0 = 0.0, cutoffGain = 0.0;
        long long pushed = 0;
        // wave 0, behind a sweep's children (and behind the root): the pushes in column order, the new top out, the next hypothesis
        // popped -- or the end of the problem (BC_STOP: the number of solutions + 1)
        auto turn = [&](int sweep, int a, bool root) {
            if (!root) {
                for (int c = a; c < M; c++) {
                    const int idx = resIdx[c];
                    if (idx < 0) continue;
                    heap_push(idx, c, resG[c]);
                    pushed++;
                }
            }
            int stop = 0;
            double gTop;
            int cur, act;
            if (bc[BC_ERR]) stop = -1;
            else if (heapN == 0) stop = sweep + 1;  // (never behind the root)
            else if (!root) {
                heap_top(gTop, cur, act);
                const double gs = emit(cur, gTop, sweep);
                if (p.useCutoff && (maximize ? (gs < gain0 - p.cutoff) : (gs > gain0 + p.cutoff))) stop = sweep + 1;  // cpp:709-719
                else if (sweep + 1 >= p.k) stop = sweep + 2;
            } else if (p.k <= 1) stop = 2;
            if (!stop) {
                heap_top(gTop, cur, act);
                heap_pop();
                if (lane == 0) { bc[BC_CUR] = cur; bc[BC_ACT] = act; }
            }
            if (lane == 0) bc[BC_STOP] = stop;
        };

        // ---- root: shortestPathCPP (cpp:119-238), N augmentations in column order on the padded problem (wave 0) ----
        if (wave == 0) {
            if (lane == 0) { bc[BC_ERR] = p.hypPerSlot < 2 ? 1 : 0; bc[BC_NEXT] = 1; }
            if (lane < D) uL[lane] = 0.0;
            wave_fence();
            bool infeasible = false;
            for (int c = 0; c < D; c++) {
                if (dijkstra<false>(Cs, LDC, uL, rl, lane, v, c4r, allRows, 0ull, c, INF, spc, pred, scanned, delta, sink)) { infeasible = true; break; }
                dual_update_flip(uL, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, c);
                wave_fence();
            }
            if (infeasible) {  // kBest2D returns 0 (cpp:588-593)
                if (lane == 0) bc[BC_STOP] = 1;
            } else if (p.hypPerSlot < 2) {
                if (lane == 0) bc[BC_STOP] = -1;
            } else {
                const double rootGain = serial_gain(Cs, LDC, lane, r4c, M, gainW);
                store_hyp(0, v, c4r, r4c, bit64(__builtin_amdgcn_readlane(r4c, 0)));  // cpp:232-235
                gain0 = emit(0, rootGain, 0);
                if (lane == 0) { resG[0] = rootGain; }
                heap_push(0, 0, rootGain);
                turn(0, 0, true);
            }
        }
        __syncthreads();
        // (every wave needs the cutoff's reference point: the root's shifted gain)
        cutoffGain = maximize ? (resG[0] - p.cutoff) : (resG[0] + p.cutoff);  // cpp:680-686
        int stop = bc[BC_STOP];
        __syncthreads();  // (resG[0] is a child's slot from here on)
        for (int sweep = 1; !stop; sweep++) {  // cpp:607-634
            const int cur = bc[BC_CUR], a = bc[BC_ACT];
            // the hypothesis that is split, into registers (every wave its own copy)
            const Hyp8 hp = hyp(cur);
            double uP = 0.0, vP = 0.0;
            int c4rP = -1, r4cP = -1, fb = 0;
            if (lane < D) { uP = hp.u[lane]; vP = hp.v[lane]; c4rP = hp.c4r[lane]; r4cP = hp.r4c[lane]; fb = hp.forb[lane]; }
            const u64 forbP = __ballot(fb != 0);
            // ---- split (cpp:455-532): the children of columns a .. M-1, each fully solved; this wave's share ----
            for (int c = a + wave; c < M; c += NW) {
                const int fr = __builtin_amdgcn_readlane(r4cP, c);         // row freed: cpp:277-278
                const u64 cand = __ballot(lane < D && c4rP >= c);           // rows of columns >= c: cpp:480-488, 525-527
                const u64 forbm = (c == a) ? forbP : bit64(fr);             // cpp:490 / 510-516
                c4r = (lane == fr) ? -1 : c4rP;
                r4c = (lane == c) ? -1 : r4cP;
                v = vP;
                if (lane < D) uL[lane] = uP;
                wave_fence();
                int idx = -1;
                if (!dijkstra<false>(Cs, LDC, uL, rl, lane, v, c4r, cand, forbm, c, INF, spc, pred, scanned, delta, sink)) {  // else infeasible: cpp:496, 521
                    dual_update_flip(uL, lane, v, c4r, r4c, spc, pred, scanned, delta, sink, c);
                    wave_fence();
                    const double g = serial_gain(Cs, LDC, lane, r4c, M, gainW);
                    if (!(p.useCutoff && (maximize ? (g < cutoffGain) : (g > cutoffGain)))) {  // cutHyp, hpp:130-131
                        if (lane == 0) idx = atomicAdd(&bc[BC_NEXT], 1);
                        idx = __builtin_amdgcn_readfirstlane(idx);
                        if (idx >= p.hypPerSlot) {  // (cannot happen: at most 1 + (k - 1) M pushes, plan_exact)
                            if (lane == 0) bc[BC_ERR] = 1;
                            idx = -1;
                        } else {
                            store_hyp(idx, v, c4r, r4c, forbm | bit64(__builtin_amdgcn_readlane(r4c, c)));  // cpp:362
                            if (lane == 0) resG[c] = g;
                        }
                    }
                }
                if (lane == 0) resIdx[c] = idx;
            }
            __syncthreads();
            if (wave == 0) turn(sweep, a, false);
            __syncthreads();
            stop = bc[BC_STOP];
        }
        if (threadIdx.x == 0) {
            p.nf[b] = stop < 0 ? -4 : stop - 1;
            if (p.pushed) p.pushed[b] = pushed;
        }
        __syncthreads();
    }
}

hipError_t launch_kbest_exact(const ExactParams &p, int grid, hipStream_t stream)
{
    if (p.B <= 0) return hipSuccess;
    if (p.maxRow <= 64) {
        // Eight waves per problem (a sweep's children in parallel) where a sweep has children to deal out -- 16 columns and more -- or
        // the batch leaves most of the chip empty anyway; one wave per problem for large batches of small frames, whose sweeps have
        // a handful of children and would pay the two barriers per sweep for nothing (1 000 integer 28 x 10 frames: 9.0 against 9.8 ms;
        // 1 024 x 64x64: 61 against 34 ms; one 64x64 problem: 36 against 9.7 ms).  KBEST_EXACT_WAVES=1 / 8 forces either (A/B).
        const int ldc = p.maxRow | 1;
        auto bytes = [&](int nw) { return (long long)(nw * 136 + 64 + 32 + 8 + ((nw + 1) & ~1) + ((p.maxRow * ldc + 1) & ~1)) * 8; };
        const char *force = getenv("KBEST_EXACT_WAVES");
        const bool many = force ? atoi(force) > 1 : (p.maxCol >= 16 || p.B < 512);
        if (many) {
            const long long base = bytes(8), withHeap = base + 16ll * p.hypPerSlot;
            if (withHeap <= 48 * 1024) hipLaunchKernelGGL((kbest_exact64_kernel<true, 8>), dim3(grid), dim3(512), (int)withHeap, stream, p);
            else hipLaunchKernelGGL((kbest_exact64_kernel<false, 8>), dim3(grid), dim3(512), (int)base, stream, p);
        } else {
            const long long base = bytes(1), withHeap = base + 16ll * p.hypPerSlot;
            if (withHeap <= 40 * 1024) hipLaunchKernelGGL((kbest_exact64_kernel<true, 1>), dim3(grid), dim3(64), (int)withHeap, stream, p);
            else hipLaunchKernelGGL((kbest_exact64_kernel<false, 1>), dim3(grid), dim3(64), (int)base, stream, p);
        }
        return hipGetLastError();
    }
    if (p.maxRow <= EXACT_LDS_ROWS) {
        // Eight waves per problem: all of them share the rows of the root's Dijkstra steps; as many of them as a CU's LDS holds scratch +
        // hypothesis for (44 bytes per row and wave, beside the shared 25 bytes per row and the sweep's results) solve a sweep's
        // children side by side: eight while two workgroups still fit a CU, else four, two, one.  KBEST_EXACT_WAVES = 1 / 2 / 4 / 8
        // forces the number of child-solving waves (A/B).
        constexpr int NW = 8;
        const long long scrB = (19ll * p.maxRow + 63) & ~63ll, hypB = (25ll * p.maxRow + 63) & ~63ll;
        auto bytes = [&](int cw) { return cw * (scrB + hypB) + hypB + 8ll * ((p.maxCol + 1) & ~1) + 4ll * ((p.maxCol + 3) & ~3) + 64 + 2 * NW * 12ll + 64; };
        const char *force = getenv("KBEST_EXACT_WAVES");
        int cw = force ? atoi(force) : (bytes(8) <= 80 * 1024 ? 8 : 4);
        cw = cw >= 8 ? 8 : (cw >= 4 ? 4 : (cw >= 2 ? 2 : 1));
        while (cw > 1 && bytes(cw) > 150 * 1024) cw >>= 1;
        const int lds = (int)bytes(cw);
        if (lds > 64 * 1024) {  // (more than the default limit of dynamic LDS)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_exactN_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
        }
        ExactParams q = p;
        q.childWaves = cw;
        hipLaunchKernelGGL(kbest_exactN_kernel<NW>, dim3(grid), dim3(64 * NW), lds, stream, q);
    } else {
        hipLaunchKernelGGL(kbest_exact_kernel, dim3(grid), dim3(64), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace kb
