// kbest_lap.h -- device code shared by the LDS kernels (kbest_engine.hip, kbest_lane.hip): one shortest augmenting
// path with lane = row (the hand-written step loop), the dual update + path flip, the serial gain, and the cycle
// stamps of diagnostic builds.  Device code only.
#ifndef KBEST_LAP_H
#define KBEST_LAP_H

#include "kbest_wave.h"

namespace kb {

// ------------------------------------------------------------ one augmentation
// Shortest augmenting path from column `start` (lane = row).  Restates the
// do{}while of shortestPathCPP (cpp:168-226) / shortestPathUpdateCPP
// (cpp:297-356):
//   cand    rows still in Row2Scan (bit r)
//   forb    rows skipped while the start column itself is scanned (cpp:310)
//   c4r     this lane's row -> column, -1 = unassigned (a sink)
//   u       LDS array, duals per column; v this lane's row dual
//   spc     shortestPathCost of this lane's row (out: valid for scanned rows)
// The strict '<' update (cpp:185, 314) is one fp64 compare.  The arg-min
// (cpp:191-194, 320-323: first minimum in ascending row order) runs on an
// order-preserving integer key of the candidates' spc: one 6-stage DPP min
// chain over the high words, ballot of the lanes that hold that minimum, and
// -- only if more than one lane ties on the high word (values within 2^-20 of
// each other, or +inf) -- a second chain over the low words; ff1 of the ballot
// picks the lowest row.  delta is then read back from the winning lane, so it
// is the exact double.  Early termination compares key high words in SALU and
// falls back to an fp64 compare only when they are equal.
// Returns 0 = path found, 1 = infeasible (cpp:197, 327), 2 = abandoned because
// delta exceeds `bound` (only when EARLY; deltaOut is then the settled distance
// at which the search was given up: a lower bound of the child's distance).
template <bool EARLY>
__device__ __forceinline__ int dijkstra(const double *Cs, int LDC, const double *u, int rl, int lane,
                                        double v, int c4r, u64 cand, u64 forb, int start, double bound,
                                        double &spc, int &pred, u64 &scannedOut, double &deltaOut,
                                        int &sinkOut, double minIn = 0.0, int sinkRow = 0, int parkFrom = 64)
{
    // parkFrom < 64 (children of a rectangular problem): rows on the zero-padded columns parkFrom .. D-1 ("parked") all
    // carry the same dual, and so do those columns, in every dual-feasible solution -- once the search has settled ONE
    // parked row at distance d every other parked row is at distance d too and scanning their columns changes nothing
    // (kbest_small.hip, file header).  The loop leaves when the chosen row's column is >= parkThr; the first time all
    // parked rows are settled with it, then parkThr is lifted.
    int parkThr = __builtin_amdgcn_readfirstlane(parkFrom);
    cand = uni64(cand);
    u64 act = cand & ~uni64(forb);
    const u64 cand0 = cand;
    int cur = uni32(start);
    // the bound is computed from LDS values (VGPRs): make it provably uniform or the loop turns divergent
    bound = __hiloint2double(uni32(__double2hiint(bound)), uni32(__double2loint(bound)));
    // EARLY with minIn > 0: the only unassigned row of a child problem is the row it freed (sinkRow), so the path
    // must END with an arc into that row, and minIn is a lower bound of the reduced cost of every such arc.  The
    // final distance is therefore at least min(spc[sinkRow], delta + minIn): the loop runs against the tighter
    // bound - minIn, and when that is reached the child is given up unless the sink is already within the bound
    // through a scanned column (then the loop goes on against the plain bound).
    minIn = __hiloint2double(uni32(__double2hiint(minIn)), uni32(__double2loint(minIn)));
    const double tight = bound - minIn;
    bool useTight = EARLY && minIn > 0.0;
    int bndHi;
    u32 bndLo;
    to_key(EARLY ? (useTight ? tight : bound) : d_inf(), bndHi, bndLo);  // bndHi <= KEY_INF_HI: one compare catches "+inf" too
    u64 dbits = 0;         // delta (0.0) as a bit pattern: it lives in a scalar register pair
    double sp = d_inf();  // this row's shortestPathCost
    int closest = 0, khi = 0, status = 0;
    u64 eq = 0;
    pred = 0;
    // LDS byte addresses (the low word of a flat LDS address is the LDS offset)
    const u32 rowAddr = (u32)reinterpret_cast<uintptr_t>(Cs + rl);
    const u32 uBase = (u32)__builtin_amdgcn_readfirstlane((int)(u32)reinterpret_cast<uintptr_t>(u));
    const int ldc8 = __builtin_amdgcn_readfirstlane(LDC * 8);
    const int keyInf = KEY_INF_HI;
    // The step loop, hand-written: at six waves per SIMD the kernel runs at the SIMD's aggregate issue rate, so every
    // instruction of this loop counts.  21 VALU + 13 SALU + 2 LDS, ONE exit test:
    //   * reduced costs are non-negative up to rounding, and for non-negative doubles the high word itself is the
    //     order-preserving key: no key conversion, and delta's high word IS the wave minimum (s81);
    //   * one step in five has several rows on the same high word, nearly always exact zeros (tight arcs): if no
    //     other of them has a smaller low word than the first, the first is the minimum (equal values: lowest row,
    //     cpp:191, 320) -- one more compare, inside the loop;
    //   * the loop goes on while (sink not reached) and (minimum key < bound key): sign bit of (mhi - bndHi) & ~cc;
    //   * the truly rare cases -- a negative candidate (-1e-17 from rounding), a high-word tie that needs the low
    //     words reduced -- leave the loop in mid-step (status 1) and are finished below in C++, which re-enters.
    // Fixed registers (all caller-saved in the AMDGPU calling convention, v20-v39: the kernel makes one call,
    // apriori_threshold, and values that live across it want the callee-saved ones): v[24:25] spc, v26 pred, v27 key, s[80:81] delta, s82 cur / col4row of the chosen row, s83
    // chosen row, s[84:85] rows still to scan, s[86:87] rows scanned against this column, s[88:89] rows at the minimum.
    // Wait states (gfx950): VALU write -> DPP read 2, VALU VGPR write -> readlane 1, VALU SGPR write -> VALU read 2.
    for (;;) {
        asm volatile(
            "L_step%=:\n\t"
            "s_mul_i32 s94, s82, s91\n\t"
            "s_lshl3_add_u32 s95, s82, s92\n\t"
            "v_add_u32_e32 v36, s94, v20\n\t"
            "v_mov_b32_e32 v39, s95\n\t"
            "ds_read_b64 v[32:33], v36\n\t"
            "ds_read_b64 v[34:35], v39\n\t"
            "v_mov_b32_e32 v37, s82\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_f64 v[30:31], s[80:81], v[32:33]\n\t"
            "v_add_f64 v[30:31], v[30:31], -v[34:35]\n\t"
            "v_add_f64 v[30:31], v[30:31], -v[22:23]\n\t"
            "v_cmp_lt_f64_e32 vcc, v[30:31], v[24:25]\n\t"
            "s_and_b64 vcc, vcc, s[86:87]\n\t"
            "v_cndmask_b32_e32 v25, v25, v31, vcc\n\t"
            "v_cndmask_b32_e64 v27, v38, v25, s[86:87]\n\t"
            "v_cndmask_b32_e32 v24, v24, v30, vcc\n\t"
            "v_cndmask_b32_e32 v26, v26, v37, vcc\n\t"
            "v_min_i32_dpp v28, v27, v27 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp v28, v28, v28 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp v28, v28, v28 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp v28, v28, v28 row_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp v28, v28, v28 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp v28, v28, v28 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_readlane_b32 s81, v28, 63\n\t"
            "s_nop 1\n\t"
            "v_cmp_eq_u32_e64 s[88:89], s81, v27\n\t"
            "s_cmp_lt_i32 s81, 0\n\t"
            "s_cbranch_scc1 L_slow%=\n\t"
            "s_ff1_i32_b64 s83, s[88:89]\n\t"
            "s_bcnt1_i32_b64 s94, s[88:89]\n\t"
            "v_readlane_b32 s80, v24, s83\n\t"
            "s_cmp_gt_u32 s94, 1\n\t"
            "s_cbranch_scc1 L_tie%=\n\t"
            "L_tail%=:\n\t"
            "s_sub_i32 s94, s81, s93\n\t"
            "v_readlane_b32 s82, v21, s83\n\t"
            "s_bitset0_b64 s[84:85], s83\n\t"
            "s_mov_b64 s[86:87], s[84:85]\n\t"
            "s_andn2_b32 s94, s94, s82\n\t"
            "s_sub_i32 s97, s82, s96\n\t"
            "s_and_b32 s94, s94, s97\n\t"
            "s_cmp_lt_i32 s94, 0\n\t"
            "s_cbranch_scc1 L_step%=\n\t"
            "s_mov_b32 s95, 0\n\t"
            "s_branch L_done%=\n\t"
            "L_tie%=:\n\t"
            "s_nop 1\n\t"
            "v_cmp_lt_u32_e64 s[76:77], v24, s80\n\t"
            "s_and_b64 s[76:77], s[76:77], s[88:89]\n\t"
            "s_cmp_eq_u64 s[76:77], 0\n\t"
            "s_cbranch_scc1 L_tail%=\n\t"
            "L_slow%=:\n\t"
            "s_mov_b32 s95, 1\n\t"
            "L_done%=:\n\t"
            : "+{s[80:81]}"(dbits), "+{s82}"(cur), "+{s[84:85]}"(cand), "+{s[86:87]}"(act), "+{v[24:25]}"(sp),
              "+{v26}"(pred), "={v27}"(khi), "={s83}"(closest), "={s[88:89]}"(eq), "={s95}"(status)
            : "{v20}"(rowAddr), "{v21}"(c4r), "{v[22:23]}"(v), "{v38}"(keyInf), "{s91}"(ldc8), "{s92}"(uBase),
              "{s93}"(bndHi), "{s96}"(parkThr)
            : "v28", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v39", "s76", "s77", "s94", "s97", "vcc", "scc",
              "memory");
        // (the compiler does not know that outputs bound to physical scalar registers are wave-uniform)
        dbits = uni64(dbits);
        cur = uni32(cur);
        cand = uni64(cand);
        act = uni64(act);
        closest = uni32(closest);
        status = uni32(status);
        int mhi = (int)(u32)(dbits >> 32);  // the wave minimum's key (s81)
        if (__builtin_expect(status != 0, 0)) {
            // finish the step here: the selects are done, khi / mhi / eq hold the high-word minimum
            const int slo = __double2loint(sp), shi = __double2hiint(sp);
            eq = uni64(eq);
            if (mhi < 0) {  // some candidate is negative: redo with the real order-preserving key
                const int sg = shi >> 31;
                khi = sel32(act, shi ^ (int)((u32)sg >> 1), KEY_INF_HI);
                mhi = wave_min_i32(khi);
                eq = __ballot(khi == mhi);
                if (__popcll(eq) > 1) {
                    const u32 t = (u32)sel32(eq, slo ^ sg, -1);
                    const u32 mlo = wave_min_u32(t);
                    eq &= __ballot(t == mlo);
                }
                closest = __builtin_ctzll(eq);
                dbits = ((u64)(u32)__builtin_amdgcn_readlane(shi, closest) << 32) | (u32)__builtin_amdgcn_readlane(slo, closest);
            } else {  // rows on the same high word, and the first of them is not the smallest: reduce the low words
                const u32 t = (u32)sel32(eq, slo, -1);
                const u32 mlo = wave_min_u32(t);
                closest = __builtin_ctzll(eq & __ballot(t == mlo));  // lowest row index: cpp:191, 320
                dbits = ((u64)(u32)mhi << 32) | mlo;
            }
            cand &= ~(1ull << closest);
            act = cand;
            cur = __builtin_amdgcn_readlane(c4r, closest);
            if (((mhi - bndHi) & ~cur & (cur - parkThr)) < 0) continue;  // not a sink, not parked, below the bound: next step
        }
        if (__builtin_expect(mhi >= bndHi, 0)) {
            const double delta = __longlong_as_double((long long)dbits);
            if (mhi >= KEY_INF_HI) { scannedOut = cand0 & ~cand; return 1; }  // minimum is +inf: infeasible (cpp:197, 327)
            if (EARLY && delta > bound) { scannedOut = cand0 & ~cand; deltaOut = delta; return 2; }  // beyond the k best (deltaOut: the distance reached)
            if (EARLY && useTight && delta > tight) {
                const int slo = __double2loint(sp), shi = __double2hiint(sp);
                const double sfr = __hiloint2double(__builtin_amdgcn_readlane(shi, sinkRow), __builtin_amdgcn_readlane(slo, sinkRow));
                if (sfr > bound) { scannedOut = cand0 & ~cand; deltaOut = delta; return 2; }  // the sink cannot come within the bound
                useTight = false;
                to_key(bound, bndHi, bndLo);
            }
        }
        if (cur < 0) break;
        if (cur >= parkThr) {  // the first parked row is settled: so are all of them, at this distance
            const u64 pk = __ballot(c4r >= parkThr) & cand;
            sp = __hiloint2double(sel32(pk, (int)(u32)(dbits >> 32), __double2hiint(sp)), sel32(pk, (int)(u32)dbits, __double2loint(sp)));
            pred = sel32(pk, cur, pred);
            cand &= ~pk;
            act = cand;
            parkThr = 64;
        }
    }
    sinkOut = closest;
    spc = sp;
    scannedOut = cand0 & ~cand;
    deltaOut = __longlong_as_double((long long)dbits);
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

// calcGain (cpp:59-80): serial left-to-right fp64 sum over the M real columns, from 0.0.  Every lane fetches its
// column's term and parks it in the wave's LDS scratch line; the chain of adds then reads the terms back as
// broadcast 16-byte reads (two terms per LDS instruction, no lane reads on the vector unit), in the reference's
// order.  Lanes >= M contribute +0.0, and x + 0.0 == x exactly for the non-negative partial sums here.
// `orig` (64-row kernel with a column order of its own, kbest_engine.hip): lane is a POSITION in that order and
// orig[lane] the reference's column index -- the terms are parked at their reference index, so the chain adds them in the
// reference's order whatever order the enumeration works in.
__device__ __forceinline__ double serial_gain(const double *Cs, int LDC, int lane, int r4c, int M, double *scratch,
                                              const unsigned char *orig = nullptr)
{
    double t = 0.0;
    if (lane < M) t = Cs[r4c + lane * LDC];
    scratch[orig ? (int)orig[lane] : lane] = t;
    wave_fence();
    double acc = 0.0;
    const double2 *terms = reinterpret_cast<const double2 *>(scratch);
    for (int j0 = 0; j0 < M; j0 += 8) {
        const double2 a = terms[(j0 >> 1)], b = terms[(j0 >> 1) + 1], c = terms[(j0 >> 1) + 2], d = terms[(j0 >> 1) + 3];
        acc = acc + a.x;
        acc = acc + a.y;
        acc = acc + b.x;
        acc = acc + b.y;
        acc = acc + c.x;
        acc = acc + c.y;
        acc = acc + d.x;
        acc = acc + d.y;
    }
    wave_fence();
    return acc;
}

// ------------------------------------------------------------ the column keys of the enumeration's own order
// Key of column c = the cost of taking its row away from it with nothing else fixed (kbest_engine.hip, phase 1b).  All the keys
// at once: in the graph whose nodes are the columns and whose arc i -> j costs what column i pays for the row that j holds (its
// reduced cost, >= 0), the key of c is the shortest cycle through c -- the diagonal of the all-pairs closure with an empty
// diagonal to start from.  Floyd-Warshall in fp32 (the keys only ORDER the columns; any order gives the same results), BLOCKED:
// 16 x 16 blocks; per block round r the diagonal block (one wave, 16 steps in lock step, no workgroup barrier), then the blocks
// of its row and column (a wave each, the same 16 steps against the finished diagonal block), then all other blocks (every
// thread, 16 independent min-plus terms per entry): three barriers per block round -- 12 for 64 columns where the plain form
// had 64.  dm: LDS scratch of Dp * Dp floats, Dp = D rounded up to 16, 16-byte aligned.  Every thread of the workgroup calls.
template <int NW>
__device__ __forceinline__ void column_keys_closure(float *dm, double *key, const double *Cs, int LDC, const double *u, const double *v,
                                                    const unsigned char *r4c, int D, int M)
{
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dp = (D + 15) & ~15, nbk = Dp >> 4;
    const float FINF = __int_as_float(0x7f800000);
    for (int e = tid; e < Dp * Dp; e += NT) {
        const int i = e / Dp, j = e - i * Dp;
        float wf = FINF;  // (the diagonal and the padding of the last block)
        if (i < D && j < D && i != j) {
            const int row = r4c[j];
            double w = (Cs[row + i * LDC] - u[i]) - v[row];
            w = w < 0.0 ? 0.0 : w;
            wf = (float)w;
        }
        dm[e] = wf;
    }
    __syncthreads();
    // one 16 x 16 block by one wave: entry (il, jl .. jl + 3) per lane; 16 steps, the wave in lock step.  A = the target's rows in
    // the columns of block r, B = block r's rows in the target's columns; whichever of them lies in the target itself changes from
    // step to step (the write back + fence), the other is final.
    const int il = lane >> 2, jl = (lane & 3) * 4;
    auto fw_block = [&](int ti, int tj, int o) {
        float *trow = dm + (ti + il) * Dp + tj + jl;
        float4 d4 = *reinterpret_cast<const float4 *>(trow);
#pragma unroll 4
        for (int kk = 0; kk < 16; kk++) {
            const float av = dm[(ti + il) * Dp + o + kk];
            const float4 b4 = *reinterpret_cast<const float4 *>(dm + (o + kk) * Dp + tj + jl);
            d4.x = fminf(d4.x, av + b4.x);
            d4.y = fminf(d4.y, av + b4.y);
            d4.z = fminf(d4.z, av + b4.z);
            d4.w = fminf(d4.w, av + b4.w);
            wave_fence();
            *reinterpret_cast<float4 *>(trow) = d4;
            wave_fence();
        }
    };
    for (int r = 0; r < nbk; r++) {
        const int o = r * 16;
        if (wave == 0) fw_block(o, o, o);
        __syncthreads();
        for (int bi = wave; bi < 2 * (nbk - 1); bi += NW) {
            const bool rowBlk = bi < nbk - 1;
            int c = rowBlk ? bi : bi - (nbk - 1);
            c += (c >= r) ? 1 : 0;
            fw_block(rowBlk ? o : c * 16, rowBlk ? c * 16 : o, o);
        }
        __syncthreads();
        const int cnt = (nbk - 1) * (nbk - 1) * 256;
        for (int e = tid; e < cnt; e += NT) {
            const int blk = e >> 8;
            int ba = blk / (nbk - 1), bb = blk - ba * (nbk - 1);
            ba += (ba >= r) ? 1 : 0;
            bb += (bb >= r) ? 1 : 0;
            const int i = ba * 16 + ((e >> 4) & 15), j = bb * 16 + (e & 15);
            float acc = dm[i * Dp + j];
#pragma unroll 4
            for (int kk = 0; kk < 16; kk++) acc = fminf(acc, dm[i * Dp + o + kk] + dm[(o + kk) * Dp + j]);
            dm[i * Dp + j] = acc;
        }
        __syncthreads();
    }
    for (int c = tid; c < M; c += NT) key[c] = (double)dm[c * Dp + c];
}

// ---- optional in-kernel cycle stamps (diagnostic builds only: make PROFILE=1).  In the shipped kernel no
// stamp executes; the stamp values only ever go to the separate `prof` buffer.
#ifdef KB_PROFILE
#define KB_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define KB_ACC(slot, expr) do { profAcc[slot] += (unsigned long long)(expr); } while (0)
#else
#define KB_T(var) do { } while (0)
#define KB_ACC(slot, expr) do { } while (0)
#endif

}  // namespace kb
#endif
