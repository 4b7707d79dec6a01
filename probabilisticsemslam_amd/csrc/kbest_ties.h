// kbest_ties.h -- ONE order of exact ties for every kernel (device code only).
//
// Hypotheses with exactly equal gain have no defined relative order in the reference: what comes out of kBest2D is an
// artefact of std::priority_queue's binary heap (shortestPathCPP.cpp:30-42, 574).  Every kernel of this engine enumerates in
// an order of its own (pool order, (parent, column), bucket order, ...), so without a rule the SAME problem could come back
// differently depending on which kernel the batch it travels in is routed to.  The rule, applied by all of them:
//
//   * solutions are ordered by (gain, row4col), row4col compared lexicographically in the REFERENCE's column order
//     (distinct assignments differ in some column, so the order is total);
//   * when the k-th and the (k+1)-th best gains are equal the k best are not a unique set: the kernel says so
//     (TIE_FLAG_BOUNDARY) and the entry point either completes the gain level and keeps the lexicographically first
//     assignments of it, or hands the flag to the caller (include/kbest_c.h, "Order of exact ties").
//
// The enumeration kernels get there with two small changes: they enumerate ONE solution more than asked for (its gain is all
// that is kept: measured free, tests/dev/kplus1.py) and, when their tables are complete, one wave runs tie_tail() below over
// the problem's own tables: a scan of the gains (the common, tie-free case ends there) and a rank sort of every run of equal
// gains.  The exhaustive kernel and the bounded walk see every assignment up to the k-th gain's bucket anyway: their rank sorts
// compare (gain, row4col) directly.
#ifndef KBEST_TIES_H
#define KBEST_TIES_H

#include <hip/hip_runtime.h>

#include "kbest_wave.h"

namespace kb {

constexpr int TIE_RUN_CAP = 4096;  // longest run of equal gains that tie_tail() orders (4 x u16 of LDS scratch per entry + TIE_SCR_EXTRA)
constexpr int TIE_RANK_MAX = 48;   // runs up to this length: every entry counts the entries before it (L^2 comparisons, no set-up);
                                   // longer runs: a radix sort over the columns (below)
constexpr int TIE_SCR_EXTRA = 64 * 32;  // u16 counters of the radix sort: one row of 32 digit counts per lane
constexpr int TIE_SCR_U16 = 4 * TIE_RUN_CAP + TIE_SCR_EXTRA;  // what finish_tables_kernel gives tie_tail

// Loads that do not go through the CU's L1: the tables are written by OTHER waves of the workgroup (visible in L2 after the
// barrier), and a line of the gain table is shared with the neighbouring problem, whose workgroup may run on this CU later and
// must not find our copy of the line.
__device__ __forceinline__ double tie_ld_gain(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int tie_ld_index(const int *table, long long i, bool i8)
{
    if (i8) return (int)__hip_atomic_load(reinterpret_cast<const signed char *>(table) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(table + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true iff assignment a comes before assignment b: the first column (reference order) on which their rows differ decides
__device__ __forceinline__ bool tie_lex_less(const int *row4col, long long a, long long b, int M, bool i8)
{
    for (int c = 0; c < M; c++) {
        const int ra = tie_ld_index(row4col, a + c, i8), rb = tie_ld_index(row4col, b + c, i8);
        if (ra != rb) return ra < rb;
    }
    return false;
}

// One wave (all 64 lanes) of the launch behind the enumeration (finish_tables_kernel, kbest_merge.hip).  gain points at slot 0 of ONE problem's
// gains; row4col / col4row are the launch's tables (int32, or int8 when i8), r4cBase / c4rBase the ELEMENT index of the
// problem's slot 0 in them (col4row may be null); nf slots are filled; scr: >= 4 * scrEntries + TIE_SCR_EXTRA u16 of LDS nobody
// else uses any more.  haveExtra / extra: the gain of the (nf+1)-th solution, when the kernel enumerated it (nf == the caller's k then).
// order = false: the tables lie in HOST memory (the kernels of a host-buffer entry write them there over the link): runs of equal
// gains are only REPORTED (KBEST_TIE_INSIDE) -- the entry brings them into the order on the host, where the tables are anyway
// (order_ties_host, kbest_capi.cpp); swapping rows of host memory from here costs a link round trip per access, and a soak of
// round 6 saw a table come back corrupted that way.
// Returns KBEST_TIE_* flags (wave-uniform).
__device__ __attribute__((noinline)) int tie_tail(const double *gain, int *row4col, long long r4cBase, int *col4row, long long c4rBase, int nf,
                                                  int M, int N, int ldCol, int ldRow, bool i8, unsigned short *scr, int scrEntries,
                                                  bool haveExtra, double extra, bool order = true)
{
    const int lane = threadIdx.x & 63;
    int flags = 0;
    if (nf < 1) return 0;
    // (the scan itself: plain loads -- the tables were written by the launch BEFORE this one, and nothing of this problem has
    //  been moved yet)
    if (haveExtra && gain[nf - 1] == extra) flags |= KBEST_TIE_BOUNDARY;
    bool any = false;
    for (int s = lane; s + 1 < nf; s += 64) any = any || (gain[s] == gain[s + 1]);
    if (__ballot(any) == 0ull) return flags;  // no two equal gains: what every tie-free problem pays
    flags |= KBEST_TIE_INSIDE;
    if (!order) return flags;
    const int cap = scrEntries < TIE_RUN_CAP ? scrEntries : TIE_RUN_CAP;
    unsigned short *inv = scr, *pos = scr + cap, *at = scr + 2 * cap, *xtra = scr + 3 * cap;  // (xtra: cap + TIE_SCR_EXTRA entries)
    int s = 0;
    while (s + 1 < nf) {
        // end of the run of gains equal to slot s (uniform)
        const double g = tie_ld_gain(gain + s);
        int e = s + 1;
        for (;;) {
            const int i = e + lane;
            const u64 diff = __ballot(i >= nf || !(tie_ld_gain(gain + (i < nf ? i : nf - 1)) == g));
            if (diff) { e += __builtin_ctzll(diff); break; }
            e += 64;
        }
        const int L = e - s;
        if (L > 1 && L <= cap) {
            if (L <= TIE_RANK_MAX) {
                // rank of every entry of the run among the run (row4col lexicographic; distinct assignments: a permutation)
                for (int i = lane; i < L; i += 64) {
                    int rank = 0;
                    for (int j = 0; j < L; j++) rank += (j != i && tie_lex_less(row4col, r4cBase + (long long)(s + j) * ldCol, r4cBase + (long long)(s + i) * ldCol, M, i8)) ? 1 : 0;
                    inv[rank] = (unsigned short)i;
                }
            } else {
                // A long run (integer-like costs: hundreds to thousands of equal gains; L^2 lexicographic comparisons of Murty
                // neighbours, which share long prefixes, took tens of milliseconds): LSD radix sort of the run's indices over the
                // columns, last column first, two 5-bit digits per column (indices < 1 024), each pass a stable counting sort -- a
                // lane owns a contiguous chunk of the current order and a private row of 32 counters, so nothing is atomic and
                // equal keys keep their order.  4 096 entries x 64 columns: ~1 ms.
                unsigned short *pa = inv, *pb = xtra, *cnt = xtra + cap;
                const int chunk = (L + 63) >> 6, lo = lane * chunk < L ? lane * chunk : L, hi = lo + chunk < L ? lo + chunk : L;
                for (int i = lane; i < L; i += 64) pa[i] = (unsigned short)i;
                wave_fence();
                // (the loops are kept rolled: unrolled, this rare path cost the finishing launch 248 registers)
#pragma unroll 1
                for (int c = M - 1; c >= 0; c--) {
#pragma unroll 1
                    for (int shift = 0; shift < 10; shift += 5) {
#pragma unroll 1
                        for (int d = 0; d < 32; d++) cnt[lane * 32 + d] = 0;
                        int seen = 0;
#pragma unroll 1
                        for (int i = lo; i < hi; i++) {
                            const int key = tie_ld_index(row4col, r4cBase + (long long)(s + pa[i]) * ldCol + c, i8);
                            cnt[lane * 32 + ((key >> shift) & 31)]++;
                            seen |= key;
                        }
                        wave_fence();
                        // where the entries of (digit, lane) start: all smaller digits, then the same digit on the lanes before
                        int tot = 0;
                        if (lane < 32) {
#pragma unroll 4
                            for (int l2 = 0; l2 < 64; l2++) tot += cnt[l2 * 32 + lane];
                        }
                        int incl = tot;  // (lanes 32 .. 63 carry zeros: the scan runs over the whole wave)
                        for (int dd = 1; dd < 32; dd <<= 1) {
                            const int t = __shfl_up(incl, dd);
                            if (lane >= dd) incl += t;
                        }
                        const int base = incl - tot;
                        const bool oneDigit = __ballot(lane < 32 && tot == L) != 0ull;  // every key has the same digit: the order stands
                        if (!oneDigit) {
                            if (lane < 32) {
                                int run = base;
#pragma unroll 1
                                for (int l2 = 0; l2 < 64; l2++) {
                                    const int n = cnt[l2 * 32 + lane];
                                    cnt[l2 * 32 + lane] = (unsigned short)run;
                                    run += n;
                                }
                            }
                            wave_fence();
#pragma unroll 1
                            for (int i = lo; i < hi; i++) {
                                const int idx = pa[i];
                                const int key = tie_ld_index(row4col, r4cBase + (long long)(s + idx) * ldCol + c, i8);
                                const int d = (key >> shift) & 31;
                                pb[cnt[lane * 32 + d]++] = (unsigned short)idx;
                            }
                            wave_fence();
                            unsigned short *t = pa; pa = pb; pb = t;
                        }
                        // (the high digit only where some key has bits there: rows below 32 need one pass per column)
                        if (__ballot((seen >> 5) != 0) == 0ull) break;
                    }
                }
                if (pa != inv) {
                    for (int i = lane; i < L; i += 64) inv[i] = pa[i];
                }
            }
            wave_fence();
            for (int i = lane; i < L; i += 64) {
                pos[i] = (unsigned short)i;
                at[i] = (unsigned short)i;
            }
            wave_fence();
            // in place, by swaps: position q of the run receives the entry of rank q
            for (int q = 0; q < L; q++) {
                const int en = inv[q], cur = pos[en];
                if (cur != q) {
                    const long long a = r4cBase + (long long)(s + q) * ldCol, b = r4cBase + (long long)(s + cur) * ldCol;
                    for (int c = lane; c < M; c += 64) {
                        const int x = tie_ld_index(row4col, a + c, i8), y = tie_ld_index(row4col, b + c, i8);
                        put_index(row4col, a + c, y, i8);
                        put_index(row4col, b + c, x, i8);
                    }
                    if (col4row) {
                        const long long a2 = c4rBase + (long long)(s + q) * ldRow, b2 = c4rBase + (long long)(s + cur) * ldRow;
                        for (int r = lane; r < N; r += 64) {
                            const int x = tie_ld_index(col4row, a2 + r, i8), y = tie_ld_index(col4row, b2 + r, i8);
                            put_index(col4row, a2 + r, y, i8);
                            put_index(col4row, b2 + r, x, i8);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the next swap may read what this one wrote
                    if (lane == 0) {
                        const int other = at[q];
                        at[cur] = (unsigned short)other;
                        pos[other] = (unsigned short)cur;
                        at[q] = (unsigned short)en;
                        pos[en] = (unsigned short)q;
                    }
                    wave_fence();
                }
            }
        } else if (L > cap) {
            flags |= KBEST_TIE_UNORDERED;
        }
        s = e;
    }
    return flags;
}

}  // namespace kb
#endif
