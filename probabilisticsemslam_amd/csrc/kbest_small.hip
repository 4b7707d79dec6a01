// kbest_small.hip -- MI355X (gfx950) k-best assignment kernel for problems of up to 32 rows, and the fused
// association path (cost block in -> probabilities out in ONE launch) built on it.
//
// The reference's production caller (getAssignmentProbs, assignment.cpp:38-74, once per frame from
// system.cpp:268) solves (nL + nM) x nM problems with a few dozen rows: conditionCosts (assignment.cpp:439-525) ->
// kBest2DCutoff(k, cutoff 42) (shortestPathCPP.cpp:646-733) -> exp-weights (assignment.cpp:616-648).  For these the
// 64-row kernel of kbest_engine.hip leaves half of every wavefront idle and pays five workgroup barriers per round.
// This kernel is organised around them instead:
//
//   * half-wave workers: a wavefront is two independent 32-lane workers (lane & 31 = row).  The Dijkstra step of
//     shortestPathUpdateCPP (cpp:307-325) runs for two children at once in a hand-written loop (dijkstra2): the DPP
//     min-reductions stop at 32 lanes, the row sets of both children share one 64-bit scalar mask (low / high word),
//     everything that is uniform within a half (column, distance, chosen row, bound) lives in scalar registers;
//   * implicit zero columns: the reference pads an N x M problem to N x N with zero columns (cpp:582-585) so that
//     inherited duals stay valid when a child frees a row.  Dual feasibility forces every row on a padded column
//     ("parked") to carry the same v, and all padded columns the same u = -v: once the search has settled ONE parked
//     row at distance d, every other parked row is at distance d too and scanning their columns changes nothing.
//     The kernel therefore keeps the padded columns implicit -- when the first parked row is settled all parked rows
//     are settled with it and ONE relaxation through an all-zero tile column (index maxCol; its dual, -v of the parked
//     rows, sits in a private slot of the worker) stands for all their columns.  The root is solved on the rectangular
//     problem (M augmentations instead of N: the reference's root needs ~N^2/2 Dijkstra steps on a 28 x 10 frame
//     because of the ties on the zero columns), children take 3-4 steps instead of ~10.  In exact arithmetic this is
//     the same shortest-path computation; the assignments, their order and the gains -- re-summed in the reference's
//     column order from the cost matrix (calcGain, cpp:59-80) -- are identical; only the internal dual variables
//     differ in the last bits, and col4row numbers the parked rows M, M+1, ... in ascending row order (SURVEY 8(a)
//     quirk 6: values >= M are "padded", not compared);
//   * rounds with three barriers and no control block: per round every worker takes one not-yet-split candidate of
//     the sorted pool (found by ballot walks, redundantly per wave), brings its saved hypothesis into its LDS node
//     block and filters its children by their first-step minimum (on square problems also by the last-arc bound);
//     barrier; all waves draw the surviving children in pairs from one list (one child per half; the parent is
//     whichever node block the entry names), solve them, keep the completed ones in full and append them to the fresh
//     list; barrier; rank merge of the old pool and the fresh list into the other pool buffer; barrier.  Emission
//     bookkeeping is recomputed by every wave from the pool order, so nothing serial sits between the merge and the
//     next split;
//   * all completed children are kept in full (no lazy re-solve): hypothesis states live in HBM slots drawn from an
//     LDS free list, slots of candidates that drop out of the pool are recycled, so 2k + (children of one round)
//     slots are enough;
//   * fused association: with `condition` the workgroup runs conditionCosts on the raw block while it loads the
//     tile; with `weights` the epilogue accumulates exp(best - g) per emitted solution in the reference's order
//     (assignment.cpp:620-640), normalises and scatters back to the original landmark numbering (:68-74): only
//     [nM][nL+1] doubles leave the kernel.
//
// fp64 add / sub / compare (and exp in the weights) only -- no MFMA.  Compiled without fast-math.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "kbest_engine.h"
#include "kbest_wave.h"

namespace kb {

namespace {

constexpr int SM_PARKED = 64;  // col4row value of a row that sits on one of the (implicit) zero columns
constexpr u64 SM_LO = 0x00000000FFFFFFFFull, SM_HI = 0xFFFFFFFF00000000ull;
constexpr u32 SM_SPLIT = 0x80000000u;  // pool meta: children already generated
constexpr double SM_GATE = 42.0;       // assignment.cpp:9

// value for the lower half-wave, value for the upper one -> one lane value
__device__ __forceinline__ int pick(int forLow, int forHigh) { return sel32(SM_HI, forHigh, forLow); }

// min over each 32-lane half, returned in every lane of that half.  EXEC must be all ones.
#define KB_HALF_MIN_CHAIN(OP)                                                          \
    "s_nop 1\n\t" OP " %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"            \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"          \
    "s_nop 1\n\t"

__device__ __forceinline__ int half_min_i32(int x)
{
    int t;
    asm volatile(KB_HALF_MIN_CHAIN("v_min_i32_dpp") : "=&v"(t) : "v"(x));
    return pick(__builtin_amdgcn_readlane(t, 31), __builtin_amdgcn_readlane(t, 63));
}
__device__ __forceinline__ u32 half_min_u32(u32 x)
{
    u32 t;
    asm volatile(KB_HALF_MIN_CHAIN("v_min_u32_dpp") : "=&v"(t) : "v"(x));
    return (u32)pick(__builtin_amdgcn_readlane((int)t, 31), __builtin_amdgcn_readlane((int)t, 63));
}

// lane `idxLow` of the lower half / lane `idxHigh` of the upper half, broadcast to the respective half
__device__ __forceinline__ int half_bcast_i32(int x, int idxLow, int idxHigh)
{
    return pick(__builtin_amdgcn_readlane(x, idxLow & 31), __builtin_amdgcn_readlane(x, 32 + (idxHigh & 31)));
}
__device__ __forceinline__ double half_bcast_f64(double x, int idxLow, int idxHigh)
{
    return __hiloint2double(half_bcast_i32(__double2hiint(x), idxLow, idxHigh), half_bcast_i32(__double2loint(x), idxLow, idxHigh));
}

// Two shortest augmenting paths at once, one per half-wave (lane & 31 = row).  Restates the do{}while of
// shortestPathCPP (cpp:168-226) / shortestPathUpdateCPP (cpp:297-356) for each half, with the zero-padded columns
// implicit (file header).  Values named ..v are lane values that are uniform within a half.
//   Cs, LDC     cost tile (LDS), real columns only
//   uArr        this half's column duals (LDS pointer, differs between the halves)
//   v, c4r      this lane's row dual and row -> column (-1 = unassigned: a sink; SM_PARKED = on a zero column)
//   cand        rows still to scan (bit = lane); forb: rows skipped while the start column itself is scanned (cpp:310)
//   live        lanes of the halves that take part (all 32 bits of a half, or none)
//   boundv      early termination (EARLY): a half gives up as soon as its settled distance exceeds it
// Out per half: spc / pred per row, scanned rows, final distance, sink row, the parked row through which the zero
// columns were entered (or -1), status 0 = path found, 1 = infeasible (cpp:197, 327), 2 = abandoned.
#ifdef KS_TRUST_SGPR
#define KS_UNI(x) (x)
#else
#define KS_UNI(x) uni32(x)
#endif

template <bool EARLY>
__device__ __forceinline__ void dijkstra2(const double *Cs, double *uW, int hubCol, int rl, double v, int c4r, u64 cand,
                                          u64 forb, bool liveA, bool liveB, int startA, int startB, double boundA,
                                          double boundB, double &spOut, int &predOut, u64 &scannedOut, double &deltaOut,
                                          int &sinkOut, int &hubRowOut, int &statusOut)
{
    // Everything that is uniform within a half lives in SCALAR registers here (suffix A = lower half, B = upper half).
    //   Cs       cost tile, column stride 33 doubles, plus one all-zero column at index hubCol
    //   uW       this half's PRIVATE copy of the column duals (LDS; differs between the halves); uW[hubCol] is the hub's
    //            dual: the zero columns are one more column of the tile as far as the step is concerned
    constexpr int LDC = 33;
    u64 liveMask = (liveA ? SM_LO : 0ull) | (liveB ? SM_HI : 0ull);
    cand = uni64(cand) & liveMask;
    const u64 cand0 = cand;
    u64 act = cand & ~uni64(forb);
    const u64 parked = __ballot(c4r == SM_PARKED);
    int curA = uni32(startA), curB = uni32(startB);
    int dHiA = 0, dLoA = 0, dHiB = 0, dLoB = 0;          // settled distance (delta) of each half, as bits
    int stA = 0, stB = 0, sinkA = 0, sinkB = 0, hubRowA = -1, hubRowB = -1;
    int finHiA = 0, finLoA = 0, finHiB = 0, finLoB = 0;  // distance at which each half ended (the loop's registers run on)
    u64 unscanned = 0ull;                                // rows an ended half never settled
    const int bHiA = uni32(__double2hiint(boundA)), bHiB = uni32(__double2hiint(boundB));
    const u32 bLoA = (u32)uni32(__double2loint(boundA)), bLoB = (u32)uni32(__double2loint(boundB));
    int spHi = KEY_INF_HI, spLo = 0, pred = 0;
    const u32 rowAddr = (u32)reinterpret_cast<uintptr_t>(Cs + rl);   // LDS byte addresses (the low word of a flat LDS
    const u32 uBase = (u32)reinterpret_cast<uintptr_t>(uW);           //   address is the LDS offset)
    const u32 hubAddr = uBase + (u32)hubCol * 8u;
    const int keyInf = KEY_INF_HI;
    while (liveMask) {
        int status, c0, c1, m0, m1, dlo0, dlo1, cnA, cnB;
        u64 eq;
        {
            // The step loop, hand-written (the kernel runs at the instruction issue rate of lone waves: every instruction
            // of this loop counts).  One pass = one Dijkstra step of BOTH halves: two LDS reads, three fp64 adds, the strict
            // '<' update of spc / pred (cpp:183-188, 313-318), a 5-stage DPP min over each 32-lane half on the raw high
            // word (reduced costs are >= 0 up to rounding: the high word is an order-preserving key), lowest row among
            // equal values (cpp:191-194, 320-323).  The loop goes on while both halves simply move to the next column;
            // everything else -- a sink, a parked row, the bound, an empty or negative minimum, a high-word tie that the
            // first row does not win, a half that is finished -- leaves it: status 0 = the step's choice is made (rows
            // c0 / c1, their keys, low words and columns are in the out registers), nothing committed; status 1 = the
            // choice itself needs the exact path.  A finished half (liveMask) takes part inertly.
            // Fixed registers: v26 +inf key, v27 u base, v28 row address, v29 c4r, v[30:31] v, v[32:33] spc, v34 pred;
            // s[70:71] live lanes, s[80:83] delta A / B, s84/s85 column A / B, s[86:87] rows to scan, s[88:89] rows to relax,
            // s[90:91] upper-half mask, s92 column stride in bytes, s93/s94 bound high words.
            int dAlo = dLoA, dAhi = dHiA, dBlo = dLoB, dBhi = dHiB;
            asm volatile(
                "L_pstep%=:\n\t"
                "v_mov_b32_e32 v44, s84\n\t"
                "v_mov_b32_e32 v45, s85\n\t"
                "v_cndmask_b32_e64 v44, v44, v45, s[90:91]\n\t"
                "v_mad_u32_u24 v46, v44, s92, v28\n\t"
                "v_lshl_add_u32 v47, v44, 3, v27\n\t"
                "ds_read_b64 v[40:41], v46\n\t"
                "ds_read_b64 v[42:43], v47\n\t"
                "v_mov_b32_e32 v48, s80\n\t"
                "v_mov_b32_e32 v45, s82\n\t"
                "v_cndmask_b32_e64 v48, v48, v45, s[90:91]\n\t"
                "v_mov_b32_e32 v49, s81\n\t"
                "v_mov_b32_e32 v45, s83\n\t"
                "v_cndmask_b32_e64 v49, v49, v45, s[90:91]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f64 v[38:39], v[48:49], v[40:41]\n\t"
                "v_add_f64 v[38:39], v[38:39], -v[42:43]\n\t"
                "v_add_f64 v[38:39], v[38:39], -v[30:31]\n\t"
                "v_cmp_lt_f64_e32 vcc, v[38:39], v[32:33]\n\t"
                "s_and_b64 vcc, vcc, s[88:89]\n\t"
                "v_cndmask_b32_e32 v33, v33, v39, vcc\n\t"
                "v_cndmask_b32_e32 v32, v32, v38, vcc\n\t"
                "v_cndmask_b32_e32 v34, v34, v44, vcc\n\t"
                "s_and_b64 s[76:77], s[86:87], s[70:71]\n\t"
                "v_cndmask_b32_e64 v35, v26, v33, s[76:77]\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp v36, v35, v35 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp v36, v36, v36 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp v36, v36, v36 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp v36, v36, v36 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp v36, v36, v36 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_readlane_b32 s72, v36, 31\n\t"
                "v_readlane_b32 s73, v36, 63\n\t"
                "s_nop 0\n\t"
                "v_mov_b32_e32 v37, s72\n\t"
                "v_mov_b32_e32 v45, s73\n\t"
                "v_cndmask_b32_e64 v37, v37, v45, s[90:91]\n\t"
                "v_cmp_eq_u32_e64 s[98:99], v37, v35\n\t"
                "s_and_b64 s[98:99], s[98:99], s[76:77]\n\t"
                "s_or_b32 s76, s72, s73\n\t"
                "s_cmp_lt_i32 s76, 0\n\t"
                "s_cbranch_scc1 L_pslow%=\n\t"
                // first row of each half at the minimum (0 for a half without one), and "a LIVE half has none"
                "s_ff1_i32_b32 s96, s98\n\t"
                "s_ff1_i32_b32 s97, s99\n\t"
                "s_cmp_eq_u32 s98, 0\n\t"
                "s_cselect_b32 s96, 0, s96\n\t"
                "s_cselect_b32 s76, s70, 0\n\t"
                "s_cmp_eq_u32 s99, 0\n\t"
                "s_cselect_b32 s97, 0, s97\n\t"
                "s_cselect_b32 s77, s71, 0\n\t"
                "s_or_b32 s76, s76, s77\n\t"
                "s_add_u32 s97, s97, 32\n\t"
                "s_cmp_lg_u32 s76, 0\n\t"
                "s_cbranch_scc1 L_pslow%=\n\t"
                "v_readlane_b32 s74, v32, s96\n\t"
                "v_readlane_b32 s75, v32, s97\n\t"
                "s_bcnt1_i32_b32 s76, s98\n\t"
                "s_bcnt1_i32_b32 s77, s99\n\t"
                "s_max_u32 s76, s76, s77\n\t"
                "s_cmp_gt_u32 s76, 1\n\t"
                "s_cbranch_scc1 L_ptie%=\n\t"
                "L_pchoice%=:\n\t"
                "v_readlane_b32 s78, v29, s96\n\t"
                "v_readlane_b32 s79, v29, s97\n\t"
                // the chosen row of a live half is PARKED (on a zero column) and its key is below the bound: enter the zero
                // columns here -- all parked rows of the half are settled at this distance, the hub's dual slot becomes
                // -v[row], the next column is the hub column
                "s_cmp_eq_u32 s78, 64\n\t"
                "s_cbranch_scc0 L_pnoparkA%=\n\t"
                "s_cmp_eq_u32 s70, 0\n\t"
                "s_cbranch_scc1 L_pnoparkA%=\n\t"
                "s_cmp_ge_i32 s72, s93\n\t"
                "s_cbranch_scc1 L_pnoparkA%=\n\t"
                "v_readlane_b32 s76, v30, s96\n\t"
                "v_readlane_b32 s77, v31, s96\n\t"
                "s_xor_b32 s77, s77, 0x80000000\n\t"
                "s_mov_b64 vcc, exec\n\t"
                "s_mov_b64 exec, 1\n\t"
                "v_mov_b32_e32 v46, s76\n\t"
                "v_mov_b32_e32 v47, s77\n\t"
                "ds_write_b64 v25, v[46:47]\n\t"
                "s_mov_b64 exec, vcc\n\t"
                "s_and_b32 s76, s68, s86\n\t"
                "s_mov_b32 s77, 0\n\t"
                "v_mov_b32_e32 v45, s72\n\t"
                "v_cndmask_b32_e64 v33, v33, v45, s[76:77]\n\t"
                "v_mov_b32_e32 v45, s74\n\t"
                "v_cndmask_b32_e64 v32, v32, v45, s[76:77]\n\t"
                "s_andn2_b32 s86, s86, s76\n\t"
                "s_mov_b32 s66, s96\n\t"
                "s_mov_b32 s78, s67\n\t"
                "L_pnoparkA%=:\n\t"
                "s_cmp_eq_u32 s79, 64\n\t"
                "s_cbranch_scc0 L_pnoparkB%=\n\t"
                "s_cmp_eq_u32 s71, 0\n\t"
                "s_cbranch_scc1 L_pnoparkB%=\n\t"
                "s_cmp_ge_i32 s73, s94\n\t"
                "s_cbranch_scc1 L_pnoparkB%=\n\t"
                "v_readlane_b32 s76, v30, s97\n\t"
                "v_readlane_b32 s77, v31, s97\n\t"
                "s_xor_b32 s77, s77, 0x80000000\n\t"
                "s_mov_b64 vcc, exec\n\t"
                "s_mov_b32 exec_lo, 0\n\t"
                "s_mov_b32 exec_hi, 1\n\t"
                "v_mov_b32_e32 v46, s76\n\t"
                "v_mov_b32_e32 v47, s77\n\t"
                "ds_write_b64 v25, v[46:47]\n\t"
                "s_mov_b64 exec, vcc\n\t"
                "s_mov_b32 s76, 0\n\t"
                "s_and_b32 s77, s69, s87\n\t"
                "v_mov_b32_e32 v45, s73\n\t"
                "v_cndmask_b32_e64 v33, v33, v45, s[76:77]\n\t"
                "v_mov_b32_e32 v45, s75\n\t"
                "v_cndmask_b32_e64 v32, v32, v45, s[76:77]\n\t"
                "s_andn2_b32 s87, s87, s77\n\t"
                "s_sub_u32 s65, s97, 32\n\t"
                "s_mov_b32 s79, s67\n\t"
                "L_pnoparkB%=:\n\t"
                // go on iff, for each half: finished, or (key below the bound's high word, column >= 0, column < 64)
                "s_sub_i32 s76, s72, s93\n\t"
                "s_andn2_b32 s76, s76, s78\n\t"
                "s_sub_i32 s77, s78, 64\n\t"
                "s_and_b32 s76, s76, s77\n\t"
                "s_orn2_b32 s76, s76, s70\n\t"
                "s_sub_i32 s77, s73, s94\n\t"
                "s_andn2_b32 s77, s77, s79\n\t"
                "s_sub_i32 vcc_lo, s79, 64\n\t"
                "s_and_b32 s77, s77, vcc_lo\n\t"
                "s_orn2_b32 s77, s77, s71\n\t"
                "s_and_b32 s76, s76, s77\n\t"
                "s_cmp_lt_i32 s76, 0\n\t"
                "s_cbranch_scc0 L_pevent%=\n\t"
                // commit: the chosen rows leave the rows to scan, their distances and columns become current
                "s_mov_b32 s80, s74\n\t"
                "s_mov_b32 s81, s72\n\t"
                "s_mov_b32 s82, s75\n\t"
                "s_mov_b32 s83, s73\n\t"
                "s_cmp_lg_u32 s70, 0\n\t"
                "s_cselect_b32 s84, s78, 0\n\t"
                "s_cmp_lg_u32 s71, 0\n\t"
                "s_cselect_b32 s85, s79, 0\n\t"
                "s_bitset0_b64 s[86:87], s96\n\t"
                "s_bitset0_b64 s[86:87], s97\n\t"
                "s_and_b64 s[88:89], s[86:87], s[70:71]\n\t"
                "s_branch L_pstep%=\n\t"
                "L_ptie%=:\n\t"
                "v_mov_b32_e32 v45, s74\n\t"
                "v_mov_b32_e32 v46, s75\n\t"
                "v_cndmask_b32_e64 v45, v45, v46, s[90:91]\n\t"
                "v_cmp_lt_u32_e64 s[76:77], v32, v45\n\t"
                "s_and_b64 s[76:77], s[76:77], s[98:99]\n\t"
                "s_cmp_eq_u64 s[76:77], 0\n\t"
                "s_cbranch_scc1 L_pchoice%=\n\t"
                "L_pslow%=:\n\t"
                "s_mov_b32 s95, 1\n\t"
                "s_branch L_pdone%=\n\t"
                "L_pevent%=:\n\t"
                "s_mov_b32 s95, 0\n\t"
                "L_pdone%=:\n\t"
                : "+{s80}"(dAlo), "+{s81}"(dAhi), "+{s82}"(dBlo), "+{s83}"(dBhi), "+{s84}"(curA), "+{s85}"(curB),
                  "+{s[86:87]}"(cand), "+{s[88:89]}"(act), "+{v32}"(spLo), "+{v33}"(spHi), "+{v34}"(pred), "={s95}"(status),
                  "={s96}"(c0), "={s97}"(c1), "={s72}"(m0), "={s73}"(m1), "={s74}"(dlo0), "={s75}"(dlo1), "={s78}"(cnA),
                  "={s79}"(cnB), "={s[98:99]}"(eq), "+{s66}"(hubRowA), "+{s65}"(hubRowB)
                : "{v25}"(hubAddr), "{v26}"(keyInf), "{v27}"(uBase), "{v28}"(rowAddr), "{v29}"(c4r), "{v[30:31]}"(v),
                  "{s67}"(hubCol), "{s[68:69]}"(parked), "{s[70:71]}"(liveMask), "{s[90:91]}"(SM_HI), "{s92}"(LDC * 8),
                  "{s93}"(bHiA), "{s94}"(bHiB)
                : "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
                  "s76", "s77", "vcc", "scc", "memory");
            // (outputs bound to physical scalar registers: the values ARE wave-uniform; KS_UNI decides whether the compiler is
            //  told so by a readfirstlane round trip)
            dLoA = KS_UNI(dAlo); dHiA = KS_UNI(dAhi); dLoB = KS_UNI(dBlo); dHiB = KS_UNI(dBhi);
            curA = KS_UNI(curA); curB = KS_UNI(curB);
            hubRowA = KS_UNI(hubRowA); hubRowB = KS_UNI(hubRowB);
            cand = uni64(cand); act = uni64(act); eq = uni64(eq);
            status = uni32(status); c0 = KS_UNI(c0); c1 = KS_UNI(c1) - 32; m0 = KS_UNI(m0); m1 = KS_UNI(m1);
            dlo0 = KS_UNI(dlo0); dlo1 = KS_UNI(dlo1); cnA = KS_UNI(cnA); cnB = KS_UNI(cnB);
        }
        u32 e0 = (u32)eq, e1 = (u32)(eq >> 32);
        if (__builtin_expect(status != 0, 0)) {
            // the selects of this step are done; make the choice on the real order-preserving key (negative values,
            // +inf, low words)
            const u64 candL = cand & liveMask;
            int khi;
            u32 klo;
            to_key(__hiloint2double(spHi, spLo), khi, klo);
            khi = sel32(candL, khi, 0x7fffffff);
            const int mh = half_min_i32(khi);
            eq = __ballot(khi == mh) & candL;
            const u32 tl = (u32)sel32(eq, (int)klo, -1);
            const u32 ml = half_min_u32(tl);
            eq &= __ballot(tl == ml);
            e0 = (u32)eq;
            e1 = (u32)(eq >> 32);
            c0 = e0 ? __builtin_ctz(e0) : 0;
            c1 = e1 ? __builtin_ctz(e1) : 0;
            m0 = e0 ? __builtin_amdgcn_readlane(spHi, c0) : KEY_INF_HI;
            m1 = e1 ? __builtin_amdgcn_readlane(spHi, 32 + c1) : KEY_INF_HI;
            dlo0 = __builtin_amdgcn_readlane(spLo, c0);
            dlo1 = __builtin_amdgcn_readlane(spLo, 32 + c1);
            cnA = __builtin_amdgcn_readlane(c4r, c0);
            cnB = __builtin_amdgcn_readlane(c4r, 32 + c1);
        }
        // bookkeeping of the step (cpp:197-224, 327-354), per half
        u64 park = 0ull;  // halves that enter the zero columns in this step
        if (liveA) {
            cand &= ~(1ull << c0);  // the chosen row leaves Row2Scan (cpp:208-210)
            if (e0 == 0 || (m0 & 0x7fffffff) >= KEY_INF_HI) { stA = 1; liveA = false; }  // minimum is +inf: infeasible (cpp:197, 327)
            else {
                dHiA = finHiA = m0;
                dLoA = finLoA = dlo0;
                if (EARLY && (m0 > bHiA || (m0 == bHiA && (u32)dlo0 > bLoA))) { stA = 2; liveA = false; }  // beyond the bound
                else if (cnA < 0) { sinkA = c0; liveA = false; }
                else if (cnA == SM_PARKED) { hubRowA = c0; curA = hubCol; park |= SM_LO; }
                else curA = cnA;
            }
        }
        if (liveB) {
            cand &= ~(1ull << (32 + c1));
            if (e1 == 0 || (m1 & 0x7fffffff) >= KEY_INF_HI) { stB = 1; liveB = false; }
            else {
                dHiB = finHiB = m1;
                dLoB = finLoB = dlo1;
                if (EARLY && (m1 > bHiB || (m1 == bHiB && (u32)dlo1 > bLoB))) { stB = 2; liveB = false; }
                else if (cnB < 0) { sinkB = c1; liveB = false; }
                else if (cnB == SM_PARKED) { hubRowB = c1; curB = hubCol; park |= SM_HI; }
                else curB = cnB;
            }
        }
        if (park) {
            // the first parked row is settled: every parked row is at this distance (equal duals), and their columns all
            // offer the other rows the same reduced costs -- settle them together; ONE relaxation through "the" zero column
            // follows, whose dual is minus the (common) dual of the parked rows
            const u64 pk = parked & cand & park;
            spHi = sel32(pk, pick(dHiA, dHiB), spHi);
            spLo = sel32(pk, pick(dLoA, dLoB), spLo);
            cand &= ~pk;
            const double vk = half_bcast_f64(v, hubRowA, hubRowB);
            if ((__lane_id() & 31u) == 0 && ((park >> __lane_id()) & 1ull)) uW[hubCol] = -vk;
            wave_fence();
        }
        liveMask = (liveA ? SM_LO : 0ull) | (liveB ? SM_HI : 0ull);
        // a half that has ended takes its rows out of the loop's registers (the loop's commit clears "the chosen row" of
        // both halves; for an ended half that must not touch anything)
        const u64 ended = ~liveMask & cand;
        unscanned |= ended;
        cand &= liveMask;
        act = cand;
    }
    spOut = __hiloint2double(spHi, spLo);
    predOut = pred;
    scannedOut = cand0 & ~(cand | unscanned);
    deltaOut = __hiloint2double(pick(finHiA, finHiB), pick(finLoA, finLoB));
    sinkOut = pick(sinkA, sinkB);
    hubRowOut = pick(hubRowA, hubRowB);
    statusOut = pick(stA, stB);
}

// updateDualAndAugment (cpp:82-117) for the halves in `ok`: path flip sink -> start through pred (through the hub:
// the sink parks, the walk goes on from the parked row the zero columns were entered by), row duals in place,
// column duals returned (lane & 31 = column; the parent's array must stay intact for its next child).
__device__ __forceinline__ void augment2(u64 ok, int l, int startv, double sp, int pred, u64 scanned, double delta, int sinkv,
                                         int hubRowv, const double *uArr, int hubCol, int r4cP, int M, double &v, int &c4r,
                                         int &r4c, double &uNew)
{
    const int hiOff = (int)(__lane_id() & 32u);
    // duals first (they use the pre-flip column -> row map r4cP)
    const u64 myBit = 1ull << __lane_id();
    const bool sc = (scanned & myBit) != 0;
    if (sc) v = v - delta + sp;  // cpp:102-106
    {
        const int rowOfCol = (l < M && r4cP >= 0 && r4cP < 32) ? r4cP : 0;
        const double spOfRow = __hiloint2double(__shfl(__double2hiint(sp), hiOff + rowOfCol), __shfl(__double2loint(sp), hiOff + rowOfCol));
        const bool rowScanned = ((scanned >> (hiOff + rowOfCol)) & 1ull) != 0;
        double un = (l < M) ? uArr[l] : 0.0;
        if (l < M && l != startv && r4cP >= 0 && rowScanned) un = un + delta - spOfRow;  // cpp:96-99
        if (l == startv) un = un + delta;                                                // cpp:92
        uNew = un;
    }
    int rv = sinkv;
    u64 going = uni64(ok);
    for (int guard = 0; going && guard < 80; guard++) {  // cpp:108-116
        const int r0 = __builtin_amdgcn_readlane(rv, 0), r1 = __builtin_amdgcn_readlane(rv, 32);
        const int cv = half_bcast_i32(pred, r0, r1);
        const bool viaHub = cv == hubCol;
        const int q0 = __builtin_amdgcn_readlane(cv, 0), q1 = __builtin_amdgcn_readlane(cv, 32);
        const int nxt = half_bcast_i32(r4c, q0 & 31, q1 & 31);
        const bool mine = ((going >> __lane_id()) & 1ull) != 0;
        if (mine && l == rv) c4r = viaHub ? SM_PARKED : cv;
        if (mine && !viaHub && l == cv) r4c = rv;
        const u64 done = __ballot(!viaHub && cv == startv);
        rv = viaHub ? hubRowv : nxt;
        going &= ~done;
    }
}

}  // namespace

#ifdef KB_PROFILE
#define KS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define KS_ACC(slot, expr) do { profAcc[slot] += (unsigned long long)(expr); } while (0)
#else
#define KS_T(var) do { } while (0)
#define KS_ACC(slot, expr) do { } while (0)
#endif

struct SCtrl {
    double cdelta;      // CDelta * numCol (cpp:583)
    double cmax;        // largest finite shifted cost (scale of the pruning margin)
    unsigned long long cmaxBits;
    int nFresh[2];      // children appended in the current round (by round parity)
    int freeTop;        // free state slots on the stack
    int nSurv;          // children that passed the filter in this round
    int nextItem;       // work queue over them
    int status;         // 0 ok, 3 infeasible root, -2 does not fit this kernel
    int N, condL;
};
static_assert(sizeof(SCtrl) <= 96, "SCtrl must fit the LDS slot reserved by small_lds_layout");

// Six waves per SIMD (80 VGPRs; three values spilled outside the loops) instead of five: 24 wave slots per CU hold three
// 8-wave problems at once, and 8 waves per problem then beat 4 up to ~3 000 frames per launch (1 000 KITTI-like frames: 0.60 ->
// 0.54 ms; at five per SIMD only 2.5 such workgroups fit and the second generation starts late).  Workgroups with an odd number
// of waves (5, 6 per problem) only pack when there are spare slots: 4 x 5 waves on 20 slots are not placed together (measured).
#ifndef KS_WAVES_PER_EU
#define KS_WAVES_PER_EU 6
#endif
template <int NW>
// (16 waves: a lone problem on its CU, four waves per SIMD -- no reason to give up registers there)
__global__ void __launch_bounds__(NW * 64)
__attribute__((amdgpu_waves_per_eu(NW == 16 ? 4 : KS_WAVES_PER_EU, NW == 16 ? 4 : KS_WAVES_PER_EU)))
kbest_small_kernel(SmallParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NW * 64, W = 2 * NW;
    const double INF = d_inf();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l = lane & 31;
    const int half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int worker = wave * 2 + half;
    const u64 myHalf = half ? SM_HI : SM_LO;
    const int b = blockIdx.x;
    const int k = p.k;
    // (a one-frame call carries its shape in the kernel arguments: no dependent loads over PCIe before the first byte of work)
    const int M = p.imm ? p.immCol : (p.nCol ? p.nCol[b] : p.maxCol);
    const int NR = p.imm ? p.immRow : (p.nRow ? p.nRow[b] : p.maxRow);  // rows of the block as given (condition: of the RAW block)
    const bool maximize = p.maximize != 0, useCut = p.useCutoff != 0;
    const SmallLds L = small_lds_layout(p.maxRow, p.maxCol, k, NW, p.weights != 0);
    constexpr int LDC = 33;  // odd: lane = row and lane = column walks are both bank-conflict free; rows beyond N hold +inf
    const int S = p.statesPerProblem;
    double *Cs = reinterpret_cast<double *>(smem + L.offC);
    double *PG = reinterpret_cast<double *>(smem + L.offPoolG);
    u32 *PM = reinterpret_cast<u32 *>(smem + L.offPoolM);
    unsigned short *PS = reinterpret_cast<unsigned short *>(smem + L.offPoolS);
    double *FG = reinterpret_cast<double *>(smem + L.offFreshG);
    u32 *FM = reinterpret_cast<u32 *>(smem + L.offFreshM);
    unsigned short *FS = reinterpret_cast<unsigned short *>(smem + L.offFreshS);
    unsigned short *freeStack = reinterpret_cast<unsigned short *>(smem + L.offFree);
    double *EG = reinterpret_cast<double *>(smem + L.offEmitG);
    unsigned short *ES = reinterpret_cast<unsigned short *>(smem + L.offEmitS);
    double *prob = reinterpret_cast<double *>(smem + L.offProb);
    unsigned short *rowIdx = reinterpret_cast<unsigned short *>(smem + L.offRowIdx);
    double *colMin = reinterpret_cast<double *>(smem + L.offColMin);
    u64 *keepBits = reinterpret_cast<u64 *>(smem + L.offKeep);
    SCtrl *ctrl = reinterpret_cast<SCtrl *>(smem + L.offCtrl);
    // this worker's node block
    unsigned char *nodeBase = smem + L.offNodes + (size_t)worker * L.nodeStride;
    double *uArr = reinterpret_cast<double *>(nodeBase);
    double *nodeV = uArr + 32;
    unsigned char *nodeC4R = nodeBase + 512;
    unsigned char *nodeR4C = nodeBase + 544;  // (+576: scalars of the node: forbidden rows, active column, state, bound)
    double *gainW = reinterpret_cast<double *>(nodeBase + 608);  // this worker's line of gain terms
    double *uW = reinterpret_cast<double *>(nodeBase + 864);     // this worker's private column duals of the child it solves (+ hub slot)
    const int hubCol = p.maxCol;                                 // the all-zero column of the tile that stands for the padded ones
    unsigned short *surv = reinterpret_cast<unsigned short *>(smem + L.offSurv);

    const long long costBase = p.costOff ? p.costOff[b] : (long long)b * p.ldRow * p.ldCol;
    const double *Cg = p.cost + costBase;
    const long long outBase = (long long)b * p.kTab;  // (exact ties, kbest_ties.h: the tables hold kTab slots -- k, or k - 1)
    double *probOut = p.weights ? p.probs + (p.probOff ? p.probOff[b] : 0) : nullptr;
    const int nLout = p.weights ? (p.imm ? p.immL : p.nL[b]) : 0;  // landmarks in the caller's numbering

#ifdef KB_PROFILE
    unsigned long long profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long profT0 = __builtin_readcyclecounter();
#endif
    // A zero-copy call polls this host-mapped counter instead of waiting for the stream: one system-scope add per
    // workgroup, behind a system-scope fence, once everything the workgroup writes has been written.
    auto signal_done = [&]() {
        if (p.done) {
            __syncthreads();
            if (tid == 0) {
                __threadfence_system();
                __hip_atomic_fetch_add(p.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    // (behind a launch of the bounded walk, kbest_bnb.hip: only what that kernel handed back)
    if (p.onlyUnfit && p.nf[b] != -2) return;
    if (tid == 0) {
        if (p.tieGain) p.tieGain[b] = __longlong_as_double(0x7ff8000000000000LL);  // no solution behind the tables (yet)
        if (p.tieFlags) p.tieFlags[b] = p.tieBase;
    }
    // ---- shapes ---------------------------------------------------------------------------------------------
    if (M == 0 || NR == 0) {  // an empty frame (getAssignmentProbs returns an empty result, assignment.cpp:50-51)
        if (tid == 0) p.nf[b] = 0;
        signal_done();
        return;
    }
    if (M < 0 || NR < 0 || M > p.maxCol || M > SMALL_MAX_DIM || (!p.condition && (NR > p.maxRow || NR < M)) ||
        (p.condition && NR > SMALL_MAX_RAW_ROWS)) {
        if (tid == 0) p.nf[b] = (p.condition && NR > SMALL_MAX_RAW_ROWS) ? -2 : -1;
        signal_done();
        return;
    }
    if (p.weights)
        for (int i = tid; i < M * (nLout + 1); i += NT) probOut[i] = 0.0;

    // ---- phase 0: the cost tile -------------------------------------------------------------------------------
    for (int i = tid; i < (p.maxCol + 1) * LDC; i += NT) Cs[i] = (i < p.maxCol * LDC) ? INF : 0.0;  // rows beyond N never win a minimum; column maxCol: zeros
    if (tid == 0) {
        ctrl->cmaxBits = 0ull;
        ctrl->status = 0;
        ctrl->nFresh[0] = 0;
        ctrl->nFresh[1] = 0;
        ctrl->freeTop = S - 1;
        ctrl->nSurv = 0;
        ctrl->nextItem = 0;
    }
    int N;  // rows of the problem that is solved
    double cdel = 0.0;
    // The block is read two or three times below (minima, kept rows, tile).  A frame-sized block is brought into LDS
    // once (the candidate pool's space is free until the root is solved): with zero-copy host input that is ONE trip
    // over PCIe instead of three.
    const double *Cr = Cg;
    if ((long long)NR * M <= 3LL * k) {
        double *stage = PG;
        for (int i = tid; i < NR * M; i += NT) stage[i] = Cg[i];
        Cr = stage;
    }
    __syncthreads();
    if (p.condition) {
        // conditionCosts (assignment.cpp:439-525): column minima (:450-458) ...
        for (int c = wave; c < M; c += NW) {
            double m = INF;
            for (int r = lane; r < NR; r += 64) m = min_keep(m, Cr[(long long)c * NR + r]);
            m = wave_min_f64(m);
            if (lane == 0) colMin[c] = m;
        }
        __syncthreads();
        // ... a row is kept iff some entry is within 42 of its column's minimum (:462-474) ...
        const int nChunk = (NR + 63) >> 6;
        for (int ch = wave; ch < nChunk; ch += NW) {
            const int r = ch * 64 + lane;
            bool good = false;
            if (r < NR)
                for (int c = 0; c < M; c++) good = good | (Cr[(long long)c * NR + r] <= colMin[c] + SM_GATE);
            const u64 m = __ballot(good);
            if (lane == 0) keepBits[ch] = m;
        }
        __syncthreads();
        int g = 0;
        for (int ch = 0; ch < nChunk; ch++) g += __popcll(keepBits[ch]);
        if (g > p.maxRow || g < M) {  // does not fit (or undefined in the reference: size_t underflow at :60)
            if (tid == 0) p.nf[b] = -2;
            signal_done();
            return;
        }
        N = g;
        // ... kept rows are compacted in order; entries become cost - colMin, or +inf beyond the gate (:476-496)
        double cm = 0.0;
        for (int r = tid; r < NR; r += NT) {
            const int ch = r >> 6;
            const u64 word = keepBits[ch];
            if (!((word >> (r & 63)) & 1ull)) continue;
            int nr = __popcll(word & ((1ull << (r & 63)) - 1ull));
            for (int c2 = 0; c2 < ch; c2++) nr += __popcll(keepBits[c2]);
            rowIdx[nr] = (unsigned short)r;
            for (int c = 0; c < M; c++) {
                const double x = Cr[(long long)c * NR + r];
                const double val = (x <= colMin[c] + SM_GATE) ? (x - colMin[c]) : INF;
                // makeCostMatrixSafe (cpp:534-569) on the conditioned matrix: every column holds an exact zero (its
                // minimum's row is kept), all entries are >= 0, so CDelta = 0 and the shift is the identity
                Cs[nr + c * LDC] = val;
                if (val < INF && val > cm) cm = val;
            }
        }
        if (cm > 0.0) atomicMax(&ctrl->cmaxBits, (unsigned long long)__double_as_longlong(cm));
    } else {
        N = NR;
        // makeCostMatrixSafe (cpp:534-569): min of C, or of -C when maximising
        double *red = FG;
        double mn = INF;
        for (int i = tid; i < N * M; i += NT) {
            double x = Cr[i];
            x = maximize ? -x : x;
            mn = min_keep(mn, x);
        }
        mn = wave_min_f64(mn);
        if (lane == 0) red[wave] = mn;
        __syncthreads();
        mn = red[0];
        for (int w = 1; w < NW; w++) mn = min_keep(mn, red[w]);
        cdel = maximize ? -mn : mn;
        double cm = 0.0;
        for (int i = tid; i < N * M; i += NT) {
            const int c = i / N, r = i - c * N;
            const double x = Cr[i];
            double val = maximize ? (-x + cdel) : (x - cdel);  // cpp:558 / cpp:564
            if (val != val) val = INF;  // inf - inf: every comparison the reference makes with it is false, like +inf
            if (val < INF && val > cm) cm = val;
            Cs[r + c * LDC] = val;
        }
        if (cm > 0.0) atomicMax(&ctrl->cmaxBits, (unsigned long long)__double_as_longlong(cm));
    }
    // free list: slot 0 is the root
    for (int i = tid; i < S - 1; i += NT) freeStack[i] = (unsigned short)(S - 1 - i);
    __syncthreads();
    const double cmaxv = __longlong_as_double((long long)ctrl->cmaxBits);
    const int nLc = N - M;  // landmarks of the solved problem (condL of assignment.cpp:60)

    // ---- single column: assignmentProb's fast path (assignment.cpp:554-570), no enumeration --------------------
    if (p.weights && M == 1) {
        if (tid == 0) {
            double norm = 0.0;
            int cnt = 0;
            for (int i = 0; i <= nLc; i++) {
                const double c = p.condition ? Cs[i] : Cr[i];
                if (c < SM_GATE) { norm += exp(-c); cnt++; }
            }
            norm = 1.0 / norm;
            for (int i = 0; i <= nLc; i++) {
                const double c = p.condition ? Cs[i] : Cr[i];
                const double q = (c < SM_GATE) ? exp(-c) : 0.0;
                probOut[(i >= nLc) ? nLout : (p.condition ? (int)rowIdx[i] : i)] = q * norm;
            }
            p.nf[b] = cnt < p.kTab ? cnt : p.kTab;
        }
        signal_done();
        return;
    }

    // hypothesis states in HBM: u[MC] v[MR] (fp64) | row4col[MC] col4row[MR] (u8) | forbidden rows (u32), activeCol, gain
    const int MC = p.maxCol, MR = p.maxRow;
    unsigned char *stBase = p.states + (long long)b * S * p.stateStride;
    const int offV = 8 * MC, offR4C = 8 * (MC + MR), offC4R = offR4C + MC, offTail = (9 * (MC + MR) + 7) & ~7;
    // store the hypothesis the half holds (lane & 31 = column for u / r4c, = row for v / c4r) for the halves in `ok`
    auto store_state2 = [&](u64 ok, int sidv, double un, double v, int r4c, int c4r, u32 forbv, double g, int av) {
        if ((ok >> lane) & 1ull) {
            unsigned char *st = stBase + (long long)sidv * p.stateStride;
            if (l < M) {
                reinterpret_cast<double *>(st)[l] = un;
                st[offR4C + l] = (unsigned char)r4c;
            }
            if (l < N) {
                reinterpret_cast<double *>(st + offV)[l] = v;
                st[offC4R + l] = (unsigned char)c4r;
            }
            if (l == 0) {
                *reinterpret_cast<u32 *>(st + offTail) = forbv;
                *reinterpret_cast<int *>(st + offTail + 4) = av;
                *reinterpret_cast<double *>(st + offTail + 8) = g;
            }
        }
    };
    // calcGain (cpp:59-80) for both halves: serial left-to-right fp64 sum over the M columns from 0.0
    auto serial_gain2 = [&](int r4c) -> double {
        double t = 0.0;
        if (l < M && r4c >= 0) t = Cs[r4c + l * LDC];
        gainW[l] = t;
        wave_fence();
        // the chain of adds reads the terms back as 16-byte reads, eight per round trip, in the reference's order; the
        // lanes beyond M parked +0.0, and x + 0.0 == x exactly for the non-negative partial sums here
        double acc = 0.0;
        const double2 *terms = reinterpret_cast<const double2 *>(gainW);
        for (int j0 = 0; j0 < M; j0 += 8) {
            const double2 a = terms[(j0 >> 1)], bq = terms[(j0 >> 1) + 1], c = terms[(j0 >> 1) + 2], d = terms[(j0 >> 1) + 3];
            acc = acc + a.x;
            acc = acc + a.y;
            acc = acc + bq.x;
            acc = acc + bq.y;
            acc = acc + c.x;
            acc = acc + c.y;
            acc = acc + d.x;
            acc = acc + d.y;
        }
        wave_fence();
        return acc;
    };
    const int rl = l < N ? l : N - 1;
    const u32 rowsMask = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);

    KS_ACC(0, __builtin_readcyclecounter() - profT0);  // [0] tile set-up
    KS_T(tRoot);
    // ---- phase 1: root LAP on the rectangular problem, worker 0 -------------------------------------------------
    if (wave == 0) {
        if (l < 32) uArr[l] = 0.0;
        wave_fence();
        double v = 0.0, un = 0.0;
        int c4r = -1, r4c = -1;
        bool bad = false;
        // Column reduction first (Jonker-Volgenant's initialisation, as in kbest_engine.hip): u[c] = min of column c, a row
        // that is the arg-min of some column goes to the lowest such column; reduced costs stay >= 0, assigned arcs are
        // tight.  The augmentations below then start only from the columns left over.
        u32 todo = (M >= 32) ? 0xffffffffu : ((1u << M) - 1u);
        {
            int *owner = reinterpret_cast<int *>(gainW);
            if (half == 0) owner[l] = 64;
            wave_fence();
            const int cc = l < M ? l : M - 1;
            const double *Ccol = Cs + cc * LDC;
            double m = INF;
            int am = 0;
            for (int r0 = 0; r0 < N; r0 += 4) {
                double x[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = Ccol[r0 + i];  // (rows N .. 31 of the tile hold +inf)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool better = x[i] < m;  // strict '<': the lowest row among equal minima
                    m = better ? x[i] : m;
                    am = better ? r0 + i : am;
                }
            }
            const bool can = half == 0 && l < M && m < INF;
            if (can) atomicMin(&owner[am], l);
            wave_fence();
            if (half == 0) uArr[l] = can ? m : 0.0;
            r4c = (can && owner[am] == l) ? am : -1;
            const int ow = (half == 0) ? owner[l] : 64;
            c4r = (half == 0 && l < N && ow < 64) ? ow : -1;
            wave_fence();
            todo &= (u32)__ballot(half == 0 && l < M && r4c < 0);
        }
        while (todo) {
            const int c = __builtin_ctz(todo);
            todo &= todo - 1;
            double sp, delta;
            int pred, sink, hubRow, status;
            u64 scanned;
            dijkstra2<false>(Cs, uArr, hubCol, rl, v, c4r, (u64)rowsMask, 0ull, true, false, c, 0, INF, INF, sp, pred, scanned,
                             delta, sink, hubRow, status);
            if (__builtin_amdgcn_readlane(status, 0) != 0) { bad = true; break; }
            const int r4cBefore = r4c;
            augment2(SM_LO, l, c, sp, pred, scanned, delta, sink, hubRow, uArr, hubCol, r4cBefore, M, v, c4r, r4c, un);
            wave_fence();
            if (half == 0 && l < M) uArr[l] = un;
            wave_fence();
        }
        un = uArr[l];
        if (bad) {
            if (lane == 0) ctrl->status = 3;
        } else {
            if (c4r < 0) c4r = SM_PARKED;  // the free rows sit on the zero columns (all with v = 0)
            const double g = serial_gain2(r4c);
            const int r0 = __builtin_amdgcn_readlane(r4c, 0);
            store_state2(SM_LO, 0, un, v, r4c, c4r, 1u << r0, g, 0);  // cpp:235: forbiddenActiveRows[row4col[0]]
            if (lane == 0) {
                PG[0] = g;
                PM[0] = 0u;
                PS[0] = 0;
                ctrl->cdelta = cdel * (double)M;  // cpp:583
            }
        }
    }
    __syncthreads();
    if (uni32(ctrl->status) == 3) {  // infeasible: kBest2D returns 0 (cpp:588-593); assignmentProb then divides by zero
        if (tid == 0) p.nf[b] = 0;
        signal_done();
        return;
    }
    const double cdelta = ctrl->cdelta;
    KS_ACC(1, __builtin_readcyclecounter() - tRoot);  // [1] root (incl. barrier)

    // ---- phase 2: rounds ---------------------------------------------------------------------------------------
    int cur = 0;         // pool buffer in use
    int nq = 1;          // entries in it
    int E = 0;           // solutions emitted so far
    double gain0u = 0.0; // gainBest[0]
    double cutG = INF;   // workMem.cutoffGain (cpp:681/684), shifted
    int extraSlot = 0;   // 1: the slot behind the last counted one was written (cutoff break, cpp:709-719)
    for (int round = 0; round < 2 * k + 8; round++) {  // (every round emits at least one solution; the cap is a safety net)
        const double *pg = PG + cur * k;
        const u32 *pm = PM + cur * k;
        const unsigned short *ps = PS + cur * k;
        KS_T(tSel);
        KS_ACC(2, 1);  // [2] rounds
        // -- select: the first W not-yet-split candidates, in pool order; this wave's two are the (2 wave)-th and the next
        int cnt = 0, firstU = -1, idx0 = -1, idx1 = -1, idxW = -1, lastOpen = -1;
        for (int base = 0; base < nq && cnt < W; base += 64) {
            const int i = base + lane;
            const bool open = i < nq && !(pm[i] & SM_SPLIT);
            const u64 m = __ballot(open);
            if (m) {
                const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (firstU < 0) firstU = base + __builtin_ctzll(m);
                const u64 h0 = __ballot(open & (rank == 2 * wave)), h1 = __ballot(open & (rank == 2 * wave + 1));
                const u64 hW = __ballot(open & (rank == W - 1));
                if (h0) idx0 = base + __builtin_ctzll(h0);
                if (h1) idx1 = base + __builtin_ctzll(h1);
                if (hW) idxW = base + __builtin_ctzll(hW);
                lastOpen = base + 63 - __builtin_clzll(m);
                cnt += __popcll(m);
            }
        }
        const int nsel = cnt < W ? cnt : W;
        // the last selected entry (the nsel-th open one): every open entry up to it is selected
        const int lastSel = (cnt >= W) ? idxW : lastOpen;
        // -- emission (kBest2D cpp:607-634): the head goes out while it has been split; the first not yet split one
        //    is split in THIS round: it is emitted too, but ends the run (its children are not in the pool yet)
        int run = (nsel > 0) ? firstU + 1 : nq;
        if (run > k - E) run = k - E;
        if (round == 0) {
            const double g0 = pg[0];
            gain0u = maximize ? (-g0 + cdelta) : (g0 + cdelta);           // cpp:599-603
            cutG = maximize ? (g0 - p.cutoff) : (g0 + p.cutoff);           // cpp:681/684
        }
        bool cutStop = false;
        int nEmit = run;
        if (useCut) {  // cpp:709-719: the first slot beyond gainBest[0] +- cutoff is written but not counted, and ends the call
            for (int base = 0; base < run && !cutStop; base += 64) {
                const int j = base + lane;
                bool beyond = false;
                if (j < run) {
                    const double g = pg[j];
                    const double gu = maximize ? (-g + cdelta) : (g + cdelta);
                    beyond = maximize ? (gu < gain0u - p.cutoff) : (gu > gain0u + p.cutoff);
                }
                const u64 m = __ballot(beyond);
                if (m) { cutStop = true; nEmit = base + __builtin_ctzll(m); }
            }
        }
        if (wave == 0) {
            for (int j = lane; j < nEmit + (cutStop ? 1 : 0); j += 64) {
                if (E + j < k) {
                    const double g = pg[j];
                    EG[E + j] = maximize ? (-g + cdelta) : (g + cdelta);  // cpp:626-630
                    ES[E + j] = ps[j];
                }
            }
        }
        const int headNew = nEmit;
        E += nEmit;
        const bool stop = cutStop || E >= k || nsel == 0;
        if (cutStop && E < k) extraSlot = 1;
        if (stop) break;

        KS_T(tNode);
        KS_ACC(3, tNode - tSel);  // [3] select + emission
        // -- this worker's node: the saved hypothesis comes into its LDS block, its children are filtered
        const int myIdx = half ? idx1 : idx0;
        const bool haveNode = worker < nsel;
        const u64 nodeMask = __ballot(haveNode);  // whole halves
        // threshold of the pool (after this round's emission): once R candidates are waiting only children below the
        // largest of them can matter
        const int R = k - E;
        const int nOld = nq - headNew;
        double T = (nOld >= R) ? pg[headNew + R - 1] : INF;
        if (useCut && !maximize && cutG < T) T = cutG;
        const int par = round & 1;
        if (nodeMask) {
            const int sidv = haveNode ? (int)ps[myIdx] : 0;
            const unsigned char *st = stBase + (long long)sidv * p.stateStride;
            double uP = 0.0, vP = 0.0;
            int r4cP = 0, c4rP = 0;
            u32 forbP = 0;
            int aP = 0;
            double gP = 0.0;
            if (haveNode) {
                if (l < M) { uP = reinterpret_cast<const double *>(st)[l]; r4cP = st[offR4C + l]; }
                if (l < N) { vP = reinterpret_cast<const double *>(st + offV)[l]; c4rP = st[offC4R + l]; }
                forbP = *reinterpret_cast<const u32 *>(st + offTail);
                aP = *reinterpret_cast<const int *>(st + offTail + 4);
                gP = *reinterpret_cast<const double *>(st + offTail + 8);
            }
            const double boundv = (T < INF) ? (T - gP) + 1e-9 * (fabs(T) + cmaxv) : INF;
            uArr[l] = uP;
            nodeV[l] = vP;
            nodeC4R[l] = (unsigned char)c4rP;
            nodeR4C[l] = (unsigned char)r4cP;
            if (l == 0) {
                *reinterpret_cast<u32 *>(nodeBase + 584) = forbP;
                *reinterpret_cast<int *>(nodeBase + 588) = aP;
                *reinterpret_cast<int *>(nodeBase + 592) = sidv;
                *reinterpret_cast<double *>(nodeBase + 600) = boundv;
            }
            wave_fence();
            KS_T(tFil);
            KS_ACC(4, tFil - tNode);  // [4] node load
            // -- first-step filter: the minimum first-step reduced cost of each child (lane & 31 = child column - a)
            //    over its candidate rows: rows of later columns and the parked rows, minus the row the child frees
            //    itself (cpp:480-488, 510-516) -- for the child on the active column minus the accumulated forbidden
            //    rows instead (cpp:490).  Same numbers as step 1 of the child's own search ((0 + C) - u) - v.
            const int cL = aP + l;
            const int cc = cL < M ? cL : M - 1;
            const double uc = uArr[cc];
            const double *Ccol = Cs + cc * LDC;
            const bool first = cL == aP;
            const int thr = first ? aP : cL + 1;       // a candidate row sits on a column >= thr (SM_PARKED > every column)
            const u32 fmask = first ? forbP : 0u;      // ... and is not forbidden for the child on the active column
            double m = INF;
            for (int r0 = 0; r0 < N; r0 += 4) {  // four independent sets of LDS reads in flight; straight-line selects only
                double vr[4], cvv[4];
                int cr[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {  // (rows N .. 31: +inf in the tile, never the minimum)
                    vr[i] = nodeV[r0 + i];
                    cr[i] = nodeC4R[r0 + i];
                    cvv[i] = Ccol[r0 + i];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double rc = (cvv[i] - uc) - vr[i];
                    const bool valid = (cr[i] >= thr) & (((fmask >> (r0 + i)) & 1u) == 0u);
                    const double rcv = valid ? rc : INF;
                    m = rcv < m ? rcv : m;
                }
            }
            // Backward bound, the other end of the path: the only sink of child c is the row it frees, fr = row4col[c],
            // and every arc into fr comes from a later column j > c -- reduced cost (C[fr,j] - u[j]) - v[fr] -- (or from the
            // zero columns, v_parked - v[fr], see below).  First arc and last arc are different arcs of the same path and all
            // reduced costs are >= 0 up to rounding, so the child's distance is at least m + minIn: beyond the bound
            // (which carries the safety margin), no wave is spent on it.
            // (Only on square problems: with zero columns the arc from them into fr is nearly always tight, the bound
            //  buys nothing and the pass costs as much as the first one -- measured on the 28 x 10 frames: 3 % fewer
            //  children, 11 % more time.)
            double minIn = 0.0;
            if (N == M) {
                minIn = INF;
                const int fr = nodeR4C[cc];
                const double vfr = nodeV[fr];
                const double *Crow = Cs + fr;
                for (int j0 = 1; j0 < M; j0 += 4) {
                    double cin[4], uj[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int j = (j0 + i < M) ? j0 + i : M - 1;
                        cin[i] = Crow[j * LDC];
                        uj[i] = uArr[j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        double rin = (cin[i] - uj[i]) - vfr;
                        rin = rin < 0.0 ? 0.0 : rin;  // -1e-17 from rounding: no information
                        const bool ok = (j0 + i < M) & (j0 + i > cL);
                        const double rv = ok ? rin : INF;
                        minIn = rv < minIn ? rv : minIn;
                    }
                }
            }
            const bool keep = haveNode && cL < M && m < INF && !(m + minIn > boundv);
            const u64 survM = __ballot(keep);
            const u64 mine = survM & myHalf;
            int base = 0;
            if (l == 0 && mine) base = atomicAdd(&ctrl->nSurv, __popcll(mine));
            base = pick(__builtin_amdgcn_readlane(base, 0), __builtin_amdgcn_readlane(base, 32));
            if (keep) surv[base + __popcll(mine & ((1ull << lane) - 1ull))] = (unsigned short)((worker << 5) | cL);
            KS_ACC(5, __builtin_readcyclecounter() - tFil);  // [5] filter
        }
        KS_T(tBA);
        __syncthreads();
        KS_T(tCh);
        KS_ACC(9, tCh - tBA);  // [9] wait at the barrier after the filter
        // -- surviving children (shortestPathUpdateCPP cpp:240-365): every wave draws them in pairs from one list, one child
        //    per half; a child's parent is whichever node block its entry names
        const int nItems = uni32(ctrl->nSurv);
        {
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&ctrl->nextItem, 2);
            for (;;) {
                const int t = uni32(ticket);
                if (t >= nItems) break;
                if (lane == 0) ticket = atomicAdd(&ctrl->nextItem, 2);  // the next ticket is drawn while this pair is solved
                const bool hasB = t + 1 < nItems;
                const int itA = uni32((int)surv[t]), itB = uni32((int)surv[hasB ? t + 1 : t]);
                const int cA = itA & 31, cB = itB & 31;
                unsigned char *blk = smem + L.offNodes + (size_t)(pick(itA, itB) >> 5) * L.nodeStride;
                uW[l] = reinterpret_cast<const double *>(blk)[l];  // private copy of the parent's column duals (+ the hub slot)
                const double vP = reinterpret_cast<const double *>(blk + 256)[l];
                const int c4rP = blk[512 + l];
                const int r4cP = (l < M) ? (int)blk[544 + l] : -1;
                const u32 forbP = *reinterpret_cast<const u32 *>(blk + 584);
                const int aP = *reinterpret_cast<const int *>(blk + 588);
                const int sidv = *reinterpret_cast<const int *>(blk + 592);
                const double boundv = *reinterpret_cast<const double *>(blk + 600);
                const double bound0 = readlane_f64(boundv, 0), bound1 = readlane_f64(boundv, 32);
                const u64 liveH = hasB ? ~0ull : SM_LO;
                const int cv = pick(cA, cB);                   // this half's child column
                const int frv = half_bcast_i32(r4cP, cA, cB);  // row freed (cpp:277-278)
                const u64 cand = __ballot(l < N && c4rP >= cv);  // rows of columns >= c and the parked rows
                const u64 forbm = __ballot((cv == aP) ? (((forbP >> l) & 1u) != 0) : (l == frv));
                int c4r = (l == frv) ? -1 : c4rP;
                double sp, delta;
                int pred, sink, hubRow, status;
                u64 scanned;
                KS_T(tD0);
                wave_fence();
                dijkstra2<true>(Cs, uW, hubCol, rl, vP, c4r, cand, forbm, true, hasB, cA, cB, bound0, bound1, sp, pred, scanned,
                                delta, sink, hubRow, status);
                KS_T(tD1);
                KS_ACC(6, tD1 - tD0);  // [6] child dijkstra (pairs)
                KS_ACC(10, __popcll(scanned));  // [10] rows scanned (both halves)
                KS_ACC(11, 1);  // [11] child passes
                u64 ok = __ballot(status == 0) & liveH;
                if (!ok) continue;
                int r4c = (l == cv) ? -1 : r4cP;
                double vN = vP, uN;
                augment2(ok, l, cv, sp, pred, scanned, delta, sink, hubRow, uW, hubCol, r4cP, M, vN, c4r, r4c, uN);
                const double g = serial_gain2(r4c);
                if (useCut) ok &= ~__ballot(maximize ? (g < cutG) : (g > cutG));  // cutHyp, cpp:496/521
                if (!ok) continue;
                int slot = 0, pos = 0;
                if (l == 0 && ((ok >> lane) & 1ull)) {
                    const int tt = atomicAdd(&ctrl->freeTop, -1) - 1;
                    slot = freeStack[tt];
                    pos = atomicAdd(&ctrl->nFresh[par], 1);
                }
                slot = pick(__builtin_amdgcn_readlane(slot, 0), __builtin_amdgcn_readlane(slot, 32));
                pos = pick(__builtin_amdgcn_readlane(pos, 0), __builtin_amdgcn_readlane(pos, 32));
                const int rnew = half_bcast_i32(r4c, cA, cB);
                const u32 forbN = ((cv == aP) ? forbP : (1u << frv)) | (1u << rnew);  // cpp:362
                store_state2(ok, slot, uN, vN, r4c, c4r, forbN, g, cv);
                if (l == 0 && ((ok >> lane) & 1ull)) {
                    FG[pos] = g;
                    FM[pos] = ((u32)sidv << 8) | (u32)cv;
                    FS[pos] = (unsigned short)slot;
                }
                KS_ACC(7, __builtin_readcyclecounter() - tD1);  // [7] finish of completed children
            }
        }
        KS_T(tB1);
        __syncthreads();
        KS_T(tMg);
        KS_ACC(8, tMg - tB1);  // [8] wait at the barrier after the children
        // -- merge: old candidates that were not emitted + this round's children -> the other buffer, by rank; the
        //    R smallest stay.  Ties in gain: old before fresh, fresh by (parent, column).
        const int nFresh = uni32(ctrl->nFresh[par]);
        if (tid == 0) { ctrl->nFresh[par ^ 1] = 0; ctrl->nSurv = 0; ctrl->nextItem = 0; }
        double *ng = PG + (cur ^ 1) * k;
        u32 *nm = PM + (cur ^ 1) * k;
        unsigned short *nsd = PS + (cur ^ 1) * k;
        for (int i = tid; i < nOld + nFresh; i += NT) {
            const bool isOld = i < nOld;
            const int j = headNew + i, f = i - nOld;
            const double g = isOld ? pg[j] : FG[f];
            const u32 meta = isOld ? (pm[j] | ((j <= lastSel) ? SM_SPLIT : 0u)) : FM[f];
            const unsigned short sid = isOld ? ps[j] : FS[f];
            // fresh gains below g (and, for a fresh entry, equal to it): four per pair of 16-byte broadcast reads
            int below = 0, equal = 0;
            {
                const double2 *f2 = reinterpret_cast<const double2 *>(FG);
                int q = 0;
                for (; q + 4 <= nFresh; q += 4) {
                    const double2 a = f2[q >> 1], c = f2[(q >> 1) + 1];
                    below += ((a.x < g) ? 1 : 0) + ((a.y < g) ? 1 : 0) + ((c.x < g) ? 1 : 0) + ((c.y < g) ? 1 : 0);
                    equal += ((a.x == g) ? 1 : 0) + ((a.y == g) ? 1 : 0) + ((c.x == g) ? 1 : 0) + ((c.y == g) ? 1 : 0);
                }
                for (; q < nFresh; q++) {
                    const double g2 = FG[q];
                    below += (g2 < g) ? 1 : 0;
                    equal += (g2 == g) ? 1 : 0;
                }
            }
            int pos;
            if (isOld) {
                pos = i + below;  // old before fresh among equal gains
            } else {
                int lo = 0, hiB = nOld;
                while (lo < hiB) {
                    const int mid = (lo + hiB) >> 1;
                    if (pg[headNew + mid] <= g) lo = mid + 1; else hiB = mid;
                }
                pos = lo + below;
                if (__builtin_expect(equal > 1, 0))  // another fresh candidate with the same gain: order by (parent, column)
                    for (int q = 0; q < nFresh; q++) pos += (FG[q] == g && FM[q] < meta) ? 1 : 0;
            }
            if (pos < R) {
                ng[pos] = g;
                nm[pos] = meta;
                nsd[pos] = sid;
            } else {
                const int t = atomicAdd(&ctrl->freeTop, 1);  // dropped for good: recycle its state slot
                freeStack[t] = sid;
            }
        }
        nq = nOld + nFresh;
        if (nq > R) nq = R;
        cur ^= 1;
        KS_T(tB2);
        KS_ACC(12, tB2 - tMg);  // [12] merge
        __syncthreads();
        KS_ACC(13, __builtin_readcyclecounter() - tB2);  // [13] ... and after the merge
    }
    KS_T(tOut);
    __syncthreads();
    // exact ties: the solution behind the tables was enumerated for its gain only
    const int nf = E > p.kTab ? p.kTab : E;
    if (E > p.kTab && tid == 0) {
        if (p.tieGain) p.tieGain[b] = EG[p.kTab];
        if (p.tieFlags && EG[p.kTab] == EG[p.kTab - 1]) p.tieFlags[b] = p.tieBase | KBEST_TIE_BOUNDARY;
    }

    // ---- phase 3: outputs ------------------------------------------------------------------------------------------
    if (p.row4col || p.col4row || p.gain) {
        for (int s = worker; s < nf; s += W) {  // one half-wave per solution
            const unsigned char *st = stBase + (long long)ES[s] * p.stateStride;
            if (p.row4col && l < M) put_index(p.row4col, (outBase + s) * p.ldCol + l, st[offR4C + l], p.tabI8 != 0);
            if (p.col4row) {
                const int c = (l < N) ? (int)st[offC4R + l] : 0;
                const u64 pk = __ballot(l < N && c == SM_PARKED) & myHalf;
                const int rank = __popcll(pk & ((1ull << lane) - 1ull));
                if (l < N) put_index(p.col4row, (outBase + s) * p.ldRow + l, (c == SM_PARKED) ? M + rank : c, p.tabI8 != 0);
            }
        }
        if (p.gain)
            for (int s = tid; s < nf + extraSlot && s < p.kTab; s += NT) p.gain[outBase + s] = EG[s];
    }
    if (p.weights) {
        // assignmentProb's accumulation (assignment.cpp:616-648), in the reference's order: solutions ascending,
        // total and every probs[col][row] summed sequentially.  The row maps of the emitted hypotheses are gathered
        // into LDS first (one parallel pass over HBM).  Then every probs[col][row] is ONE thread's register: the thread walks the
        // solutions in order and adds the weights of those that put its row on its column -- the same additions in the same order
        // as the reference's read-modify-write of the table, without the table (a serial LDS read-modify-write per solution by
        // one wave while the others waited: 18 000 cycles of a 28 x 10 frame's 530 000).  Tables with more entries than the
        // workgroup has threads keep the walk by columns.
        unsigned char *rTab = smem + L.offNodes;                    // [nf][M] rows (the node blocks are dead now)
        double *wts = PG;                                           // [nf] weights (the pool is dead now)
        const int tabCap = (W * L.nodeStride) / (M > 0 ? M : 1);
        const double best = EG[0];
        for (int s = tid; s < nf; s += NT) {
            const double g = EG[s];
            wts[s] = (p.gate && !(best + SM_GATE > g)) ? 0.0 : exp(best - g);  // :622-626 (0.0: skipped -- x + 0.0 is x, bit for bit, for the x >= +0.0 here)
        }
        const int nAcc = M * (nLc + 1);
        const bool direct = nAcc <= NT;
        const int accC = direct && tid < nAcc ? tid / (nLc + 1) : 0, accR = tid - accC * (nLc + 1);
        double acc = 0.0;
        if (!direct)
            for (int i = tid; i < nAcc; i += NT) prob[i] = 0.0;
        double total = 0.0;
        for (int s0 = 0; s0 < nf; s0 += tabCap) {
            const int ns = (nf - s0) < tabCap ? (nf - s0) : tabCap;
            __syncthreads();
            for (int i = tid; i < ns * M; i += NT) {
                const int s = i / M, c = i - s * M;
                const int r = stBase[(long long)ES[s0 + s] * p.stateStride + offR4C + c];
                rTab[i] = (unsigned char)(r >= nLc ? nLc : r);  // (every unassigned measurement's own row counts as "no landmark": :634-637)
            }
            __syncthreads();
            if (direct && wave * 64 < nAcc) {  // (only the waves that hold entries of the table walk)
                const unsigned char *rp = rTab + accC;
                const double *wp = wts + s0;
                int s = 0;
                for (; s + 8 <= ns; s += 8) {  // (eight solutions' reads in flight; the additions stay in order)
                    double w[8];
                    int r[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        w[q] = wp[s + q];
                        r[q] = rp[(s + q) * M];
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double a2 = acc + w[q];
                        total = total + w[q];
                        acc = (r[q] == accR) ? a2 : acc;  // :633-638
                    }
                }
                for (; s < ns; s++) {
                    const double w = wp[s];
                    const double a2 = acc + w;
                    total = total + w;
                    acc = ((int)rp[s * M] == accR) ? a2 : acc;
                }
            } else if (!direct && wave == 0) {
                for (int s = 0; s < ns; s++) {
                    const double w = wts[s0 + s];
                    if (w == 0.0) continue;  // (skipped, or underflowed to nothing)
                    total += w;
                    if (lane < M) {
                        const int r = rTab[s * M + lane];
                        prob[lane * (nLc + 1) + r] += w;  // :633-638
                    }
                }
            }
        }
        if (direct) {
            if (tid < nAcc) {
                const double norm = 1.0 / total;  // :643
                // scatter back to the caller's landmark numbering (getAssignmentProbs, assignment.cpp:68-74)
                const int ro = (accR >= nLc) ? nLout : (p.condition ? (int)rowIdx[accR] : accR);
                probOut[accC * (nLout + 1) + ro] = acc * norm;
            }
        }
        __syncthreads();
        if (!direct && wave == 0) {
            const double norm = 1.0 / total;  // :643
            for (int i = lane; i < M * (nLc + 1); i += 64) {
                const int c = i / (nLc + 1), r = i - c * (nLc + 1);
                // scatter back to the caller's landmark numbering (getAssignmentProbs, assignment.cpp:68-74)
                const int ro = (r >= nLc) ? nLout : (p.condition ? (int)rowIdx[r] : r);
                probOut[c * (nLout + 1) + ro] = prob[i] * norm;
            }
        }
    }
    if (tid == 0) p.nf[b] = nf;
    signal_done();
#ifdef KB_PROFILE
    profAcc[14] = __builtin_readcyclecounter() - tOut;   // [14] outputs / weights epilogue
    profAcc[15] = __builtin_readcyclecounter() - profT0;  // [15] whole kernel (this wave)
    if (p.prof && lane == 0)
        for (int i = 0; i < 16; i++) atomicAdd(p.prof + (long long)b * 16 + i, profAcc[i]);
#endif
}

template <int NW>
static hipError_t launch_small_nw(const SmallParams &p, int B, hipStream_t stream)
{
    const SmallLds L = small_lds_layout(p.maxRow, p.maxCol, p.k, NW, p.weights != 0);
    // (raising the dynamic LDS limit is a runtime call of several microseconds: only when this launch needs more than
    //  any before it -- a one-frame call is a single launch and nothing else; per device, as the attribute is)
    static std::atomic<int> granted[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (L.total > granted[dev & 15].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kbest_small_kernel<NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L.total);
        if (e != hipSuccess) return e;
        granted[dev & 15].store(L.total, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((kbest_small_kernel<NW>), dim3(B), dim3(NW * 64), L.total, stream, p);
    return hipGetLastError();
}

hipError_t launch_kbest_small(const SmallParams &p, int B, int nWaves, hipStream_t stream)
{
    switch (nWaves) {
    case 2: return launch_small_nw<2>(p, B, stream);
    case 4: return launch_small_nw<4>(p, B, stream);
    case 8: return launch_small_nw<8>(p, B, stream);
    default: return launch_small_nw<16>(p, B, stream);
    }
}

}  // namespace kb
