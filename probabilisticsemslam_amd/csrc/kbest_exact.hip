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
// One WAVE per problem: the lanes share the rows of a Dijkstra step (row = lane, lane + 64, ...: the reduced costs
// ((delta + C) - u) - v left to right, the strict-'<' update, the minimum with the LOWEST row among equal values: cpp:183-191,
// 313-320), the dual update and the copies of a hypothesis; the queue and the path flip are one lane's.  The cost copy, the queue
// and the pool of hypotheses (25 N + 16 bytes each, one per push) live in an HBM work space; up to 1 024 rows the scratch of a
// search and the hypothesis being solved are in LDS (a child only reaches the pool if it is feasible and not cut).  This is the
// slow, total, literal path -- 256 problems of 64 x 64, k = 200 take 138 ms here (0.54 ms per problem; the reference on one host core:
// 10 ms per problem) against 0.7 ms on the LDS kernel; 1 000 integer-cost 28 x 10 problems 12 ms against 4.5 -- and is only taken
// when asked for or when no other kernel takes the size.
#include <hip/hip_runtime.h>

#include "kbest_engine.h"
#include "kbest_wave.h"

namespace kb {

namespace {

constexpr int EXACT_LDS_ROWS = 1024;  // up to here a search's scratch (19 bytes per row) and the hypothesis being solved (25) lie in LDS
constexpr int EXACT_LDS_C_ROWS = 64;  // ... and the padded cost copy as well

struct ExLayout {  // byte offsets inside one slot of the work space (D = maxRow, H = hypotheses per slot)
    long long C, spc, pred, scanCols, scanRow, inScan, forbStart, heap, freeL, hgain, pool, hypStride, total;
    // inside one hypothesis
    long long hu, hv, hc4r, hr4c, hforb, hact;
    __host__ __device__ ExLayout(long long D, long long H)
    {
        auto up = [](long long x) { return (x + 63) & ~63ll; };
        hu = 0; hv = up(8 * D); hc4r = hv + up(8 * D); hr4c = hc4r + up(4 * D); hforb = hr4c + up(4 * D); hact = hforb + up(D);
        hypStride = up(hact + 16);
        long long o = 0;
        C = o;         o += up(8 * D * D);
        spc = o;       o += up(8 * D);
        pred = o;      o += up(4 * D);
        scanCols = o;  o += up(4 * D);
        scanRow = o;   o += up(D);
        inScan = o;    o += up(D);
        forbStart = o; o += up(D);
        heap = o;      o += up(16 * H);  // (gain, hypothesis) pairs: one load per comparison
        freeL = o;     o += up(4 * H);
        hgain = o;     o += up(8 * H);
        pool = o;      o += hypStride * H;
        total = up(o);
    }
};

struct Hyp {
    double *u, *v;
    int *c4r, *r4c;
    unsigned char *forb;
    int *act;
};

}  // namespace

long long exact_slot_bytes(int maxRow, int hypPerSlot) { return ExLayout(maxRow, hypPerSlot).total; }

// MODE 0: everything in the work space (more than EXACT_LDS_ROWS rows); 1: the scratch of a search and the hypothesis being solved in
// LDS; 2: the padded cost copy as well (up to EXACT_LDS_C_ROWS rows).  A template parameter, not a run-time choice: the pointers must
// be LDS pointers at compile time (flat accesses to LDS cost a global access' latency).
template <int MODE>
__global__ void __launch_bounds__(64) kbest_exact_kernel(ExactParams p)
{
    const int lane = threadIdx.x;
    const double INF = d_inf();
    const ExLayout L(p.maxRow, p.hypPerSlot);
    unsigned char *ws = p.work + (long long)blockIdx.x * L.total;
    // the padded cost copy: in LDS up to EXACT_LDS_C_ROWS rows (32 KB at 64), else in the work space
    constexpr bool cInLds = MODE == 2;
    // the scratch of a search (ScratchSpace, hpp:73-142): in LDS up to EXACT_LDS_ROWS rows, in the work space beyond
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool inLds = MODE >= 1;
    const long long Dm = p.maxRow;
    double *spc = inLds ? reinterpret_cast<double *>(lds) : reinterpret_cast<double *>(ws + L.spc);
    int *pred = inLds ? reinterpret_cast<int *>(lds + 8 * Dm) : reinterpret_cast<int *>(ws + L.pred);
    int *scanCols = inLds ? reinterpret_cast<int *>(lds + 12 * Dm) : reinterpret_cast<int *>(ws + L.scanCols);
    unsigned char *scanRow = inLds ? lds + 16 * Dm : ws + L.scanRow;
    unsigned char *inScan = inLds ? lds + 17 * Dm : ws + L.inScan;
    unsigned char *forbStart = inLds ? lds + 18 * Dm : ws + L.forbStart;
    // ... and so does the hypothesis that is being solved (a child is copied from its parent into LDS, solved there, and only
    // goes to the pool in HBM if it is feasible and not cut): 25 bytes per row more
    const long long ldsHyp0 = (19 * Dm + 63) & ~63ll;
    const Hyp hLds{reinterpret_cast<double *>(lds + ldsHyp0), reinterpret_cast<double *>(lds + ldsHyp0 + 8 * Dm),
                   reinterpret_cast<int *>(lds + ldsHyp0 + 16 * Dm), reinterpret_cast<int *>(lds + ldsHyp0 + 20 * Dm),
                   lds + ldsHyp0 + 24 * Dm, reinterpret_cast<int *>(lds + ldsHyp0 + 25 * Dm + (8 - (25 * Dm) % 8) % 8)};
    double *Cw = cInLds ? reinterpret_cast<double *>(lds + ((ldsHyp0 + 25 * Dm + 16 + 63) & ~63ll)) : reinterpret_cast<double *>(ws + L.C);
    struct HeapE { double g; long long idx; };
    HeapE *heap = reinterpret_cast<HeapE *>(ws + L.heap);
    int *freeL = reinterpret_cast<int *>(ws + L.freeL);
    double *hgain = reinterpret_cast<double *>(ws + L.hgain);
    const bool tabI8 = (p.flags & KBEST_FLAG_TABLES_I8) != 0;
    const bool maximize = p.maximize != 0;
    __shared__ int sh[8];  // [0] heap size, [1] free count, [2] scratch

    auto hyp = [&](int i) {
        unsigned char *b = ws + L.pool + (long long)i * L.hypStride;
        return Hyp{reinterpret_cast<double *>(b + L.hu), reinterpret_cast<double *>(b + L.hv), reinterpret_cast<int *>(b + L.hc4r),
                   reinterpret_cast<int *>(b + L.hr4c), b + L.hforb, reinterpret_cast<int *>(b + L.hact)};
    };
    // what one lane wrote to the work space, every lane reads after this (one wave, one CU: program order + a fence)
    auto sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads(); };

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
        for (int i = lane; i < p.hypPerSlot; i += 64) freeL[i] = p.hypPerSlot - 1 - i;  // a stack: hypothesis 0 first
        if (lane == 0) { sh[0] = 0; sh[1] = p.hypPerSlot; }
        sync();

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
                const bool forbNow = useForb && cur == start;
                double best = INF;
                int bestR = 0x7fffffff;
                for (int r = lane; r < D; r += 64) {
                    if (!inScan[r]) continue;
                    if (forbNow && forbStart[r]) continue;
                    const double rc = ((delta + Ccol[r]) - uc) - h.v[r];  // cpp:183 / 313: left to right
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
        // std::priority_queue<pMurtyHyp> with a < b <=> a.gain > b.gain (cpp:35-37): libstdc++'s __push_heap / __adjust_heap
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
        auto heap_push = [&](int hidx) {
            if (lane == 0) {
                const int n = sh[0];
                sh[0] = n + 1;
                sift_up(n, 0, HeapE{hgain[hidx], hidx});
            }
            sync();
        };
        auto heap_pop = [&]() -> int {
            if (lane == 0) {
                const int top = (int)heap[0].idx;
                const int len = sh[0] - 1;
                const HeapE val = heap[len];
                sh[0] = len;
                if (len > 0) {
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
                sh[2] = top;
            }
            sync();
            return sh[2];
        };
        auto alloc_hyp = [&]() -> int {
            if (lane == 0) {
                const int n = sh[1];
                sh[2] = n > 0 ? freeL[n - 1] : -1;
                if (n > 0) sh[1] = n - 1;
            }
            sync();
            return sh[2];
        };
        auto free_hyp = [&](int i) {
            if (lane == 0) { freeL[sh[1]] = i; sh[1] = sh[1] + 1; }
            sync();
        };
        auto emit = [&](int hidx, int slot) {
            const Hyp h = hyp(hidx);
            for (int c = lane; c < M; c += 64) put_index(p.row4col, (outBase + slot) * p.ldCol + c, h.r4c[c], tabI8);
            if (p.col4row)
                for (int r = lane; r < N; r += 64) put_index(p.col4row, (outBase + slot) * p.ldRow + r, h.c4r[r], tabI8);
            const double g = hgain[hidx];
            const double out = maximize ? (-g + CDelta) : (g + CDelta);  // cpp:626-630
            if (lane == 0) p.gain[outBase + slot] = out;
            return out;
        };

        // ---- root: shortestPathCPP (cpp:119-238), N augmentations in column order on the padded problem ----
        const int root = alloc_hyp();
        if (root < 0) { if (lane == 0) p.nf[b] = -4; continue; }
        const Hyp hrG = hyp(root);
        const Hyp hr = inLds ? hLds : hrG;
        for (int i = lane; i < D; i += 64) { hr.c4r[i] = -1; hr.r4c[i] = -1; hr.u[i] = 0.0; hr.v[i] = 0.0; hr.forb[i] = 0; }
        if (lane == 0) hr.act[0] = 0;
        sync();
        int infeasible = 0;
        for (int c = 0; c < D && !infeasible; c++) {
            for (int r = lane; r < D; r += 64) inScan[r] = 1;
            sync();
            infeasible = augment(hr, c, false);
        }
        if (infeasible) {  // kBest2D returns 0 (cpp:588-593)
            if (lane == 0) p.nf[b] = 0;
            sync();
            continue;
        }
        // a hypothesis solved in LDS goes to its record in the pool
        auto store_hyp = [&](const Hyp &src, const Hyp &dst) {
            for (int i = lane; i < D; i += 64) { dst.r4c[i] = src.r4c[i]; dst.c4r[i] = src.c4r[i]; dst.u[i] = src.u[i]; dst.v[i] = src.v[i]; dst.forb[i] = src.forb[i]; }
            if (lane == 0) dst.act[0] = src.act[0];
            sync();
        };
        {
            const double g = gain_of(hr, M);
            if (lane == 0) { hgain[root] = g; hr.forb[hr.r4c[0]] = 1; }  // cpp:232-235
            sync();
            if (inLds) store_hyp(hr, hrG);
        }
        const double gain0 = emit(root, 0);
        const double cutoffGain = maximize ? (hgain[root] - p.cutoff) : (hgain[root] + p.cutoff);  // cpp:680-686
        heap_push(root);
        long long pushed = 0;
        int sweep = 1, err = 0;
        for (; sweep < p.k; sweep++) {  // cpp:607-634
            const int cur = heap_pop();
            const Hyp hp = hyp(cur);
            const int a = hp.act[0];
            // ---- split (cpp:455-532): the children of columns a .. M-1, each fully solved, pushed in that order ----
            for (int c = a; c < M && !err; c++) {
                // rows still owned by columns >= c of the parent (cpp:480-488; 506-508, 512, 525-527)
                for (int r = lane; r < D; r += 64) { inScan[r] = 0; forbStart[r] = (c == a) ? hp.forb[r] : 0; }  // cpp:490 / 510
                sync();
                for (int j = c + lane; j < D; j += 64) inScan[hp.r4c[j]] = 1;
                if (c != a && lane == 0) forbStart[hp.r4c[c]] = 1;  // cpp:516
                sync();
                int ch = -1;
                if (!inLds) {
                    ch = alloc_hyp();
                    if (ch < 0) { err = 1; break; }
                }
                const Hyp hc = inLds ? hLds : hyp(ch);
                for (int i = lane; i < D; i += 64) {  // shortestPathUpdateCPP, cpp:262-278
                    hc.r4c[i] = hp.r4c[i]; hc.c4r[i] = hp.c4r[i]; hc.u[i] = hp.u[i]; hc.v[i] = hp.v[i]; hc.forb[i] = forbStart[i];
                }
                sync();
                if (lane == 0) { hc.act[0] = c; hc.c4r[hc.r4c[c]] = -1; hc.r4c[c] = -1; }
                sync();
                if (augment(hc, c, true)) { if (!inLds) free_hyp(ch); continue; }  // infeasible child: gain -1, dropped (cpp:496, 521)
                const double g = gain_of(hc, M);
                const bool cut = p.useCutoff && (maximize ? (g < cutoffGain) : (g > cutoffGain));  // cutHyp, hpp:130-131
                if (cut) { if (!inLds) free_hyp(ch); continue; }
                if (lane == 0) hc.forb[hc.r4c[c]] = 1;  // cpp:362
                sync();
                if (inLds) {
                    ch = alloc_hyp();
                    if (ch < 0) { err = 1; break; }
                    store_hyp(hc, hyp(ch));
                }
                if (lane == 0) hgain[ch] = g;
                sync();
                heap_push(ch);
                pushed++;
            }
            if (err) break;
            free_hyp(cur);
            if (sh[0] == 0) break;
            const double gs = emit((int)heap[0].idx, sweep);
            if (p.useCutoff) {  // cpp:709-719
                if (!maximize) { if (gs > gain0 + p.cutoff) break; }
                else           { if (gs < gain0 - p.cutoff) break; }
            }
        }
        if (lane == 0) {
            p.nf[b] = err ? -4 : sweep;
            if (p.pushed) p.pushed[b] = pushed;
        }
        sync();
    }
}

hipError_t launch_kbest_exact(const ExactParams &p, int grid, hipStream_t stream)
{
    if (p.B <= 0) return hipSuccess;
    int lds = p.maxRow <= EXACT_LDS_ROWS ? ((19 * p.maxRow + 63) & ~63) + 25 * p.maxRow + 16 + 64 : 0;
    if (p.maxRow <= EXACT_LDS_C_ROWS) lds = ((lds + 63) & ~63) + 8 * p.maxRow * p.maxRow;
    if (p.maxRow <= EXACT_LDS_C_ROWS) hipLaunchKernelGGL(kbest_exact_kernel<2>, dim3(grid), dim3(64), lds, stream, p);
    else if (p.maxRow <= EXACT_LDS_ROWS) hipLaunchKernelGGL(kbest_exact_kernel<1>, dim3(grid), dim3(64), lds, stream, p);
    else hipLaunchKernelGGL(kbest_exact_kernel<0>, dim3(grid), dim3(64), lds, stream, p);
    return hipGetLastError();
}

}  // namespace kb
