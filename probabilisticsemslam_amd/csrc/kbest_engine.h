// kbest_engine.h -- internal interface between the C ABI (kbest_capi.cpp) and
// the device code (kbest_engine.hip).  Not installed; include/kbest_c.h is the
// public boundary.
#ifndef KBEST_ENGINE_H
#define KBEST_ENGINE_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kbest_c.h"

namespace kb {

typedef unsigned long long u64;
typedef unsigned int u32;

// Kernel arguments of one batched launch (all pointers are device pointers).
struct Params {
    const double *cost;       // packed column-major cost blocks
    const long long *costOff; // per-problem offset in doubles, or nullptr (uniform)
    const int *nRow;          // per-problem shapes, or nullptr (uniform maxRow x maxCol)
    const int *nCol;
    int maxRow, maxCol;       // capacity of this launch: larger problems get nf = -1
    int ldRow, ldCol;         // leading dimensions of the outputs / of the default cost packing (>= maxRow, maxCol)
    int k;
    int maximize, useCutoff;
    unsigned flags;
    double cutoff;
    int rootColOffset, rootColStride;
    int *row4col;             // [B][k][maxCol]
    int *col4row;             // [B][k][maxRow]
    double *gain;             // [B][k]
    int *nf;                  // [B]
    long long *pushed;        // [B] or nullptr
    unsigned char *states;    // workspace: [B][statesPerProblem] saved hypotheses, stateStride bytes each
    long long stateStride;
    int statesPerProblem;     // all state slots of one matrix: [0, lazyStates) + the eager region behind it
    int lazyStates;           // >= k: slots for the root and for candidates that are re-solved when selected
    int spec;                 // candidates re-solved / split per round (1 = the reference's order exactly)
    unsigned long long *prof; // [B][16] cycle stamps; only read by diagnostic builds (make PROFILE=1)
    unsigned short *slotSid;  // workspace: [B][slot_table_stride(k)] state slot of each output slot
    double *dualU;            // [B][ldCol] duals of the root solution per column (MurtyHyp::u), or nullptr
    double *dualV;            // [B][ldRow] ... per row (MurtyHyp::v), or nullptr
    int gainCols;             // numCol4Gain of shortestPathCPP (cpp:232); 0 = numCol
    int split, splitB;        // split > 1 (64-row kernel): `split` workgroups per matrix, splitB matrices; tables / work space per workgroup
    unsigned long long *sharedT;  // [splitB] smallest threshold any share of a matrix has published (ordered key; ~0 = none)
    // optimistic bounds (64-row kernel; kbest_engine.hip "optimistic bounds and tickets"): a node is split against the optRho-quantile
    // of the pool's candidates instead of its last one; optRho grows linearly from optRho0 to optRho1 while the first optPhi * k
    // solutions go out.  optRho0 >= 1: off.
    float optRho0, optRho1, optSlope;  // optSlope = (optRho1 - optRho0) / (optPhi * k), by the host: no loop-invariant float arithmetic in the kernel
    double optKappa;          // a ticket's re-split goes at least optKappa * (key - optimum) beyond its key
    int optMinPool;           // candidates the pool must hold before a quantile of it is used
    // exact ties (kbest_ties.h): the tables hold kTab slots per problem -- k, or k - 1: the k-th solution is then enumerated for its
    // gain only, which goes to tieGain[problem] (NaN: there is none)
    int kTab;
    double *tieGain;
    // relay (kbest_engine.hip): relayP > 1: every matrix is enumerated by relayP workgroups one after the other (grid = relayB x
    // relayP); relayBuf: [relayB] LDS images of relayStride bytes; three words per matrix, all zero between launches (the last of a
    // matrix' workgroups to leave clears them): relayClaim -- pieces claimed (a workgroup's piece is what it claims), relayFlag --
    // pieces done (15: the matrix is finished), relayGone -- workgroups that have left
    int relayP, relayB;
    int relayFirst, relayStep;  // piece j hands over once k * (relayFirst + j * relayStep) / 1024 solutions are out (the last piece runs to the end;
                                // the launch ends with a generation of LAST pieces: they should be short; scalar integer arithmetic only)
    unsigned char *relayBuf;
    long long relayStride;
    unsigned *relayFlag;
    unsigned *relayClaim, *relayGone;
};

struct CondParams {
    const double *cost;       // raw (nL+nM) x nM blocks
    const long long *costOff;
    const int *nRow, *nCol;
    double *out;              // conditioned blocks, same offsets (goodRows x nCol, column-major)
    int *goodRows;            // [B]
    int *condL;               // [B] goodRows - nCol (assignment.cpp:60), or nullptr
    int *rowIdx;              // [B][maxRow] original row of each kept row
    int maxRow;
};

struct WeightParams {
    const int *nL, *nM;
    const double *cost;
    const long long *costOff;
    const double *gain;       // [B][k]
    const int *row4col;       // [B][k][maxCol]
    const int *nf;            // [B]
    double *probs;
    const long long *probOff;
    int k, maxCol;
    const int *rowIdx;        // [B][maxRow] or nullptr: scatter back through conditionCosts' row map
    const int *nLout;         // [B] landmarks in the output numbering (with rowIdx)
    int maxRow;
    int gate;                 // 1: assignmentProb (skip solutions beyond best + 42, :622-626); 0: bruteForceProb (:918-923)
    int solveRows;            // rows of the largest solved problem (bounds nL + 1: sizes the LDS accumulator)
    int chunk, ldsAcc;        // set by launch_weights: solutions per LDS chunk; 1: the [nM][nL+1] table accumulates in LDS
    long long accBytes;
    int kUse;                 // > 0: only the first kUse solutions of a problem are weighed (its table holds k >= kUse slots)
};

// Bytes of one saved hypothesis: u[D] v[D] (fp64), row4col[D] col4row[D] (u8),
// forbidden-row mask, gain, activeCol.
__host__ __device__ inline long long state_stride(int maxRow)
{
    // u, v (fp64), row4col, col4row (u8), forbidden mask, gain, activeCol, columns whose child has been completed (u64: the
    // 64-row kernel's re-split tickets); rounded to whole 128-byte lines: neighbouring states never share a cache line
    return ((((long long)18 * maxRow + 7) & ~7LL) + 32 + 127) & ~127LL;
}

// u16 entries of one matrix's slot -> state table, rounded to whole 128-byte lines
__host__ __device__ inline long long slot_table_stride(int k) { return (((long long)k * 2 + 127) & ~127LL) / 2; }

// LDS carve-up of one workgroup (= one cost matrix).
constexpr int OPT_TICKETS = 64;   // re-split tickets a problem can hold (one wave sorts them)
constexpr int OPT_BYTES = 16 * 8 * 3 + 16 + 8 + OPT_TICKETS * 10 + 8;  // struct Opt (kbest_engine.hip)
struct Lds {
    int offC, offNodes, nodeStride, offFreshG, offFreshM, offPoolG, offPoolM, offPoolS, offSurv, offFreshS, offRootMap, offPerm, offCtrl, offOpt, offGainW, total;
};

__host__ __device__ inline Lds lds_layout(int maxRow, int k, int spec, int nWaves)
{
    Lds L;
    const int ldc = maxRow | 1;
    int o = 0;
    L.offC = o;          o += maxRow * ldc * 8;      // shifted, zero-padded cost tile
    L.nodeStride = (18 * maxRow + 24 + 7) & ~7;      // u, v (fp64), scalars, row4col, col4row (u8)
    L.offNodes = o;      o += spec * L.nodeStride;   // solved hypotheses waiting to be split
    L.offFreshG = o;     o += (spec * 64 > 16 ? spec * 64 : 16) * 8;  // surviving children of this round: gain
                                                     //   (also: the first-step minima during the filter phase)
    L.offPoolG = o;      o += k * 8;                 // sorted candidate pool: gain
    L.offFreshM = o;     o += spec * 64 * 4;         //   (parent state, column); during the filter: last-arc minima
    L.offPoolM = o;      o += k * 4;                 //   (parent state, column, flags)
    L.offPoolS = o;      o += k * 2;                 //   own state slot
    L.offSurv = o;       o += spec * 64 * 2;         // children that passed the filter: (last-arc bound / 7 bits, node, column)
    L.offFreshS = o;     o += spec * 64 * 2;         //   own state slot of the surviving children
    L.offRootMap = o;    o += maxRow;                // the optimum's col4row (u8): atoms of the a-priori threshold
    o = (o + 3) & ~3;
    L.offPerm = o;       o += 128;                   // column order of the enumeration: original column of a position, and back
    o = (o + 7) & ~7;
    L.offCtrl = o;       o += 240;                   // struct Ctrl
    o = (o + 7) & ~7;
    L.offOpt = o;        o += OPT_BYTES;             // struct Opt: optimistic bounds, per-node accumulators, re-split tickets
    o = (o + 15) & ~15;
    L.offGainW = o;      o += nWaves * 512;          // one line of gain terms per wave (calcGain)
    L.total = (o + 15) & ~15;
    return L;
}

// computeQuadricCostMatrix inputs: per frame nL landmark and nM measurement (mean[3], cov[3][3]) pairs
struct QuadricParams {
    const int *nL, *nM;
    const long long *landOff, *measOff;  // first landmark / measurement of each frame
    const double *landMean, *landCov, *measMean, *measCov;
    double gate;                         // NONASSIGN_QUADRIC
    double *cost;                        // (nL+nM) x nM column-major per frame
    const long long *costOff;
};

// computeBBCostMatrix / asgnBB inputs: boxes are (xmin, ymin, xmax, ymax, xOffset)
struct BoxParams {
    const int *nL, *nR;
    const long long *offL, *offR;
    const double *boxL, *boxR;
    double gate;                         // NONASSIGN_BOUNDBOX
    double *cost;                        // (nR+nL) x nL column-major per frame
    const long long *costOff;
    int *assign;                         // [sum nL] matched right box or -1
};

// ---- general-size kernel (kbest_wide.hip): numRow up to 64 * 16, any k; hypotheses and pool in HBM work space ----
constexpr int WIDE_NW = 8;         // waves per problem (16 in the one-workgroup-per-CU shapes, 4 beyond 512 rows: WideParams::nw)
constexpr int WIDE_MAX_DIM = 1024; // rows per problem (16 rows per lane at most)
constexpr int WIDE_MAX_SPEC = 64;  // hypotheses split per round at most
constexpr int WIDE_CTRL_BYTES = 1920;
constexpr int WIDE_SAMPLES = 1024; // LDS index of the sorted pool: every 64th gain (k up to 65 535; beyond: binary search in HBM)

struct WideParams {
    const double *cost;
    const long long *costOff;
    const int *nRow, *nCol;
    int B;                    // problems in the batch (the grid strides over them)
    int maxRow, maxCol;       // capacity of this launch (work space, LDS, rows per lane)
    int ldRow, ldCol;         // leading dimensions of the outputs / of the default cost packing
    int minRows;              // problems with fewer rows are left to the LDS kernel (mixed batches); 0 = take all
    int tile;                 // 1: the square cost copy lives in LDS (it fits), Cw is not used
    int spec;                 // hypotheses split per round (1 = the reference's order of operations exactly), <= WIDE_MAX_SPEC
    int nw;                   // waves per problem: 8 or 16 (4 beyond 512 rows: the waves' working sets are 20 bytes per row each)
    int k;
    int maximize, useCutoff;
    unsigned flags;
    double cutoff;
    int rootColOffset, rootColStride;
    int *row4col;             // [B][k][ldCol]
    int *col4row;             // [B][k][ldRow] or nullptr
    double *gain;             // [B][k]
    int *nf;                  // [B]
    long long *pushed;        // [B] or nullptr
    // work space, one slot per workgroup of the grid
    double *Cw;               // shifted, zero-padded square copy of the cost matrix
    long long cwStride;       // doubles per slot
    unsigned char *states;    // [statesPerProblem] saved hypotheses, stateStride bytes each
    long long stateStride;
    int statesPerProblem;     // k + maxCol + 2: pool + children of one sweep + the hypothesis being split
    double *poolG;            // two buffers of poolStride gains (sorted candidate pool, ping-pong)
    int *poolS;               //   ... and their state slots
    long long poolStride;
    int *freeList;            // stack of free state slots
    long long freeStride;
    unsigned long long *prof; // diagnostic builds (KB_PROFILE): [B][16] cycle sums over the waves
    int kTab;                 // exact ties (kbest_ties.h), as in Params
    double *tieGain;
    unsigned *queue;          // [0] problems taken beyond the first gridDim.x, [1] workgroups that have left (zero between launches); null: fixed stride
};

// bytes of one saved hypothesis of the general-size kernel: u[D] v[D] (fp64), row4col[D] col4row[D] (i32),
// forbidden rows (u32 per lane), gain, activeCol; whole 128-byte lines
__host__ __device__ inline long long wide_state_stride(int maxRow) { return (24LL * maxRow + 256 + 16 + 127) & ~127LL; }

// state slots at the end of a problem's work space that hold the atoms of the a-priori threshold: per column cost + rows
// moved (rows per lane words), then the root's gain
__host__ __device__ inline int wide_atom_slots(int maxRow, int maxCol)
{
    const int R = maxRow <= 64 ? 1 : (maxRow <= 128 ? 2 : (maxRow <= 256 ? 4 : (maxRow <= 512 ? 8 : 16)));
    const long long bytes = 8LL * ((long long)maxCol * (1 + R) + 1);
    return (int)((bytes + wide_state_stride(maxRow) - 1) / wide_state_stride(maxRow));
}

struct WideLds { int offWave, waveStride, offNode, nodeStride, offChildG, offChildS, offChildC, offSample, offRed, offCtrl, offPerm, offTile, total; };

// tile: keep the shifted square cost copy in LDS instead of the HBM work space (when maxRow^2 * 8 bytes fit)
// hypotheses split per round by the general-size kernel.  Measured (kernel ms at 1 / 2 / 4 / 8 per round).  Eight waves
// per problem, two or three problems per CU: 1024 x 100x20 4.1 / 3.2 / 2.7 / 2.5, 512 x 128x128 18.5 / 18.2 / 18.6 / 20.2,
// 512 x 96x96 12.0 / 11.5 / 11.7 / 12.3.  Sixteen waves, one problem per CU: 256 x 64x64 3.7 / 3.0 / 2.6 / 2.5,
// 256 x 128x128 10.5 / 9.3 / 8.8 / 9.1, 256 x 256x256 44.6 / 42.4 / 43.1 / 43.1.  With few columns a round is a handful of
// children and the rounds' latency dominates (speculation pays); with many columns one split already fills eight waves
// and speculative splits are wasted work; sixteen waves want twice the children per round.
// With k in the thousands (bruteForceProb, assignment.cpp:868) nearly every hypothesis near the pool's head is output
// sooner or later, so speculation is almost never wasted and the number of rounds is what costs (each a chain of dependent
// HBM round trips: 64 x 30x10, k = 20 000: 45 ms at 8 per round, 2 500 rounds): as many as the LDS copies allow.
__host__ __device__ inline int wide_spec(int maxCol, int nw = 8, int k = 200)
{
    // (re-scanned with the a-priori threshold in place, which makes the early rounds' speculative splits cheap: 16 waves,
    //  256 x 128x128 at 2 / 4 / 8 per round 7.5 / 6.8 / 6.7 ms, 256 x 256x256 29.4 / 28.0 / 27.9, 256 x 64x64 3.0 / 2.5 / 2.3;
    //  8 waves, 512 x 128x128 at 2 / 3 / 4 / 8: 13.2 / 13.0 / 13.0 / 19.6 -- the last one loses the second workgroup per CU)
    const int s = (nw == 16 ? 1024 : 384) / (maxCol > 0 ? maxCol : 1), lo = nw == 16 ? 2 : 1;
    if (k >= 1024) return WIDE_MAX_SPEC;  // (clamped to wide_spec_cap by the caller)
    return s >= 8 ? 8 : (s >= lo ? s : lo);
}
// capacity (LDS arrays, state slots): what KBEST_WIDE_SPEC may ask for
__host__ __device__ inline int wide_node_stride(int maxRow) { return (24 * maxRow + 256 + 16 + 15) & ~15; }  // LDS copy of a saved hypothesis
__host__ __device__ inline int wide_spec_cap(int maxCol, int maxRow)
{
    int s = 1024 / (maxCol > 0 ? maxCol : 1);
    const int fit = ((maxRow <= 64 ? 64 : 32) * 1024) / wide_node_stride(maxRow);  // the split hypotheses' LDS copies: 32 KiB at most (64 for small problems)
    s = s < fit ? s : fit;
    return s >= WIDE_MAX_SPEC ? WIDE_MAX_SPEC : (s >= 1 ? s : 1);
}

// (sized for the `spec` hypotheses actually split per round -- LDS decides how many problems a CU holds)
__host__ __device__ inline WideLds wide_lds_layout(int maxRow, int maxCol, bool tile, int nw, int spec)
{
    WideLds L;
    const int nc = spec * maxCol;                  // children of one round at most
    int o = 0;
    L.waveStride = (20 * maxRow + 15) & ~15;       // per wave: u (fp64), col4row, row4col, pred (i32)
    L.offWave = o;       o += nw * L.waveStride;
    L.nodeStride = wide_node_stride(maxRow);       // copy of a saved hypothesis (wide_state_stride's content)
    L.offNode = o;       o += spec * L.nodeStride;  // the hypotheses being split in this round
    int ncP = 1;
    while (ncP <= nc) ncP <<= 1;                   // (the gains: padded to a power of two for the merge's searches)
    L.offChildG = o;     o += ncP * 8;              // surviving children of the round: gain, state slot, (parent, column)
    L.offChildS = o;     o += nc * 4;
    L.offChildC = o;     o += nc * 4;
    o = (o + 7) & ~7;
    L.offSample = o;     o += WIDE_SAMPLES * 8;
    L.offRed = o;        o += nw * 8;
    L.offCtrl = o;       o += WIDE_CTRL_BYTES;      // struct WideCtrl
    L.offPerm = o;       o += 4 * maxRow;           // column order of the enumeration: original column of a position, and back (u16)
    o = (o + 15) & ~15;
    L.offTile = o;       if (tile) o += maxRow * maxRow * 8;
    L.total = (o + 15) & ~15;
    return L;
}

// ---- small-problem kernel (kbest_small.hip): numRow <= 32.  Half-wave workers (two children per wavefront), the
//      zero-padded columns of the reference kept implicit, two barriers per round, optional fused
//      conditionCosts prologue and assignmentProb epilogue (cost block in -> probabilities out, one launch) ----
constexpr int SMALL_MAX_DIM = 32;
constexpr int SMALL_MAX_K = 1024;
constexpr int TINY_MAX_COL = 8;        // kbest_tiny.hip: measurements per frame,
constexpr int TINY_MAX_ROW = 64;       //   rows of the raw block (landmarks + measurements),
constexpr int TINY_MAX_COUNT = 1 << 23; //   assignments of the frame in all: (nL + nM)! / nL!,
constexpr int TINY_MAX_PREFIX = 1 << 15; //  ... and prefixes of nM - 2 columns (each is decoded once by some thread),
constexpr int TINY_CAP = 1024;         //   candidates kept for the final sort
// feasible prefixes kept in LDS for the passes (1 024-thread workgroups: a lone frame on its CU; 256: batches, four per CU)
__host__ __device__ constexpr int tiny_prefix_cap(int nThreads) { return nThreads >= 1024 ? 2048 : 512; }
constexpr int BNB_MAX_COL = 16;        // kbest_bnb.hip: measurements per frame (rows of an assignment packed one byte each in 16 bytes),
constexpr int BNB_MAX_ROW = 64;        //   rows of the raw block,
// entries of each of the two frontier lists (1 024-thread workgroups: a lone frame on its CU; 256: batches)
#ifndef KB_BNB_FCAP_SMALL
#define KB_BNB_FCAP_SMALL 512
#endif
__host__ __device__ constexpr int bnb_frontier_cap(int nThreads) { return nThreads >= 1024 ? 1536 : KB_BNB_FCAP_SMALL; }
// candidates kept for the final sort (they lie in one frontier list: 28 bytes each), and the largest k that leaves them room
__host__ __device__ constexpr int bnb_cand_cap(int nThreads) { return bnb_frontier_cap(nThreads) * 32 / 28 - 4; }
__host__ __device__ constexpr int bnb_max_k(int nThreads) { return bnb_cand_cap(nThreads) - 64 < bnb_frontier_cap(nThreads) ? bnb_cand_cap(nThreads) - 64 : bnb_frontier_cap(nThreads); }
constexpr int SMALL_MAX_RAW_ROWS = 2048;  // rows of the unconditioned block (assoc mode)

struct SmallParams {
    const double *cost;       // packed column-major blocks
    const long long *costOff; // per-problem offset in doubles, or nullptr (uniform ldRow x ldCol packing)
    const int *nRow;          // rows per problem, or nullptr (uniform maxRow).  assoc mode: nL + nM of the RAW block
    const int *nCol;          // columns per problem, or nullptr (uniform maxCol)
    int maxRow, maxCol;       // solver capacity of this launch (<= 32): LDS tile, state layout
    int ldRow, ldCol;         // leading dimensions of row4col / col4row outputs and of the default cost packing
    int k;
    int maximize, useCutoff;
    double cutoff;
    int *row4col;             // [B][k][ldCol] or nullptr
    int *col4row;             // [B][k][ldRow] or nullptr
    double *gain;             // [B][k] or nullptr
    int *nf;                  // [B]; -1 shape error, -2 "does not fit this kernel" (the host re-runs it on the general path)
    unsigned char *states;    // workspace [B][statesPerProblem] x stateStride
    long long stateStride;
    int statesPerProblem;
    // association weights (assignment.cpp:547-683) fused behind the enumeration
    int weights;              // 1: write probabilities instead of / besides the assignments
    int condition;            // 1: conditionCosts first (assignment.cpp:439-525) and scatter back through its row map (:68-74)
    int gate;                 // 1: assignmentProb (solutions beyond best + 42 are skipped, :622-626); 0: bruteForceProb (:918-923)
    const int *nL;            // [B] landmarks (weights mode; nRow = nL + nM)
    double *probs;            // packed [nM][nL+1] per problem
    const long long *probOff;
    unsigned long long *prof; // diagnostic builds only
    int imm, immRow, immCol, immL;  // imm = 1 (B = 1): the shape of the one problem travels in the kernel arguments
    int tabI8;                // 1: row4col / col4row are int8 tables (KBEST_FLAG_TABLES_I8)
    int *done;                // host-mapped completion counter (zero-copy calls) or nullptr
    int bnbRow;               // kbest_bnb.hip: rows of the largest raw block of the launch (<= 64; sizes its tile)
    int onlyUnfit;            // kbest_small.hip: 1 = answer only the problems whose nf is -2 (handed back by the launch before)
    int kTab;                 // exact ties (kbest_ties.h), as in Params (the weights are those of the first kTab solutions)
    double *tieGain;
    int *tieFlags;            // fused association launches (no tables for the finishing kernel to look at): [B] KBEST_TIE_* or nullptr
    int tieBase;              // ... what a frame's flags start from (KBEST_TIE_UNCHECKED when k sits at the kernel's limit: no solution behind the k-th)
};

__host__ __device__ inline long long small_state_stride(int maxRow, int maxCol)
{
    // u[maxCol] v[maxRow] (fp64), row4col[maxCol] col4row[maxRow] (u8), forbidden-row mask (u32), activeCol, gain; whole 128-byte lines
    return ((((long long)9 * (maxRow + maxCol) + 7) & ~7LL) + 16 + 127) & ~127LL;
}

__host__ __device__ inline int small_states_per_problem(int k, int nWaves, int maxCol) { return 2 * k + 2 * nWaves * maxCol + 4; }

struct SmallLds {
    int offC, offNodes, nodeStride, offPoolG, offPoolM, offPoolS, offFreshG, offFreshM, offFreshS, offFree, offEmitG, offEmitS, offSurv,
        offProb, offRowIdx, offColMin, offKeep, offCtrl, total;
};

__host__ __device__ inline SmallLds small_lds_layout(int maxRow, int maxCol, int k, int nWaves, bool weights)
{
    SmallLds L;
    const int W = 2 * nWaves, S = small_states_per_problem(k, nWaves, maxCol);
    int o = 0;
    L.offC = o;        o += (maxCol + 1) * 33 * 8;       // cost tile: the real columns + ONE zero column (the padded ones), column stride 33
    L.nodeStride = 2 * 256 + 64 + 32 + 256 + 272;        // per worker: u[32] v[32] (fp64), col4row[32] row4col[32] (u8), scalars, gain-term line, private u[33]
    o = (o + 15) & ~15;
    L.offNodes = o;    o += W * L.nodeStride;
    L.offPoolG = o;    o += 2 * k * 8;                   // sorted candidate pool, two buffers: gain
    L.offEmitG = o;    o += k * 8;                       // gains of the emitted solutions (un-shifted)
    o = (o + 15) & ~15;
    L.offFreshG = o;   o += W * maxCol * 8;              // children completed in this round (16-byte aligned: read as double2)
    L.offProb = o;     o += weights ? maxCol * 33 * 8 : 0;  // probability accumulators [nM][condL + 1]
    L.offColMin = o;   o += weights ? maxCol * 8 : 0;
    L.offKeep = o;     o += weights ? (SMALL_MAX_RAW_ROWS / 64) * 8 : 0;  // kept-row bits of conditionCosts
    L.offPoolM = o;    o += 2 * k * 4;                   //   (parent state, column, flags)
    L.offFreshM = o;   o += W * maxCol * 4;
    L.offPoolS = o;    o += 2 * k * 2;                   //   own state slot
    L.offFreshS = o;   o += W * maxCol * 2;
    L.offEmitS = o;    o += k * 2;
    L.offSurv = o;     o += W * 32 * 2;                  // children that passed the filter: (node, column)
    L.offFree = o;     o += S * 2;                       // stack of free state slots
    L.offRowIdx = o;   o += weights ? 32 * 2 : 0;        // original row of each kept row
    o = (o + 15) & ~15;
    L.offCtrl = o;     o += 96;
    L.total = (o + 15) & ~15;
    return L;
}

// ---- lane-per-child kernel (kbest_lane.hip): numRow <= 32, dense batches.  One LANE per child of Murty's partition: the
//      rows of a child's Dijkstra are an unrolled loop over registers (spc[DR], pred[DR] per lane), 64 children advance one
//      step per pass; uses kb::Params (lazyStates, slotSid as in the 64-row kernel; every completed child is kept in full,
//      state slots come from an LDS free list) ----
constexpr int LANE_MAX_DIM = 32;
constexpr int LANE_MAX_SPEC = 16;

__host__ __device__ inline int lane_rows(int maxRow) { return maxRow <= 16 ? 16 : 32; }  // rows unrolled per lane (template DR)
// saved hypothesis of the lane kernel: the 64-row kernel's layout with DR rows -- u[DR] v[DR] (fp64), row4col[DR] col4row[DR]
// (u8), forbidden mask (u64), gain, activeCol; whole 128-byte lines
__host__ __device__ inline long long lane_state_stride(int maxRow) { return state_stride(lane_rows(maxRow)); }
// emitted + pool (<= k together) + the children completed in one round (<= spec * maxCol) + the root
__host__ __device__ inline int lane_states_per_problem(int k, int spec, int maxCol) { return k + spec * maxCol + 2; }

struct LaneLds {
    int offC, offNodes, nodeStride, offPoolG, offFreshG, offPoolM, offFreshM, offPoolS, offFreshS, offItems, offComp, offFree, offScr,
        scrStride, offCtrl, offGainW, total;
};
// node block (a hypothesis being split), DR = lane_rows: u[DR] v[DR] (fp64) | row4col[DR] col4row[DR] (u8) | cand[DR] (u32:
// rows of the columns >= c) | gain, bound (fp64) | forbidden mask, activeCol, state slot (i32)
__host__ __device__ inline int lane_node_bytes(int DR) { return 22 * DR + 32; }

__host__ __device__ inline LaneLds lane_lds_layout(int maxRow, int maxCol, int k, int spec, int nWaves, int lanesPerChild)
{
    LaneLds L;
    const int DR = lane_rows(maxRow), ldc = maxRow | 1, nc = spec * maxCol;
    int o = 0;
    L.offC = o;        o += maxRow * ldc * 8;            // shifted, zero-padded cost tile (column stride odd: lanes on different columns hit different banks)
    o = (o + 15) & ~15;
    // lanes of one wave read v[r] of DIFFERENT nodes in one instruction: a stride of 8 (mod 128) bytes puts them on different banks
    L.nodeStride = ((lane_node_bytes(DR) + 119) & ~127) + 8;
    if (L.nodeStride < lane_node_bytes(DR)) L.nodeStride += 128;
    L.offNodes = o;    o += spec * L.nodeStride;
    o = (o + 15) & ~15;
    L.offPoolG = o;    o += k * 8;                       // sorted candidate pool: gain
    o = (o + 15) & ~15;
    L.offFreshG = o;   o += (nc > 16 ? nc : 16) * 8;     // children completed in this round (16-byte aligned: read as double2; phase 0's scratch)
    L.offPoolM = o;    o += k * 4;                       //   (parent state, column, flags)
    L.offFreshM = o;   o += nc * 4;
    L.offPoolS = o;    o += k * 2;                       //   own state slot
    L.offFreshS = o;   o += nc * 2;
    L.offItems = o;    o += nc * 2;                      // children of this round: (node, column)
    L.offComp = o;     o += nc * 2;                      // state slots of the children that reached their sink this round
    L.offFree = o;     o += lane_states_per_problem(k, spec, maxCol) * 2;  // stack of free state slots
    o = (o + 3) & ~3;
    L.scrStride = 4 * DR + 4;                            // per finishing child: row4col, col4row (u8 each), (pred, pred's row) pairs
    L.offScr = o;      o += nWaves * (64 / lanesPerChild) * L.scrStride;  // completed children are finished 64 / lanesPerChild at a time per wave
    o = (o + 15) & ~15;
    L.offCtrl = o;     o += 160;
    L.offGainW = o;    o += 512;                         // wave 0: the root's scratch line
    L.total = (o + 15) & ~15;
    return L;
}

// k-way merge of per-shard k-best lists into the global k best (kbest_merge.hip).  The lists of shard s start
// s * shardStride BYTES behind those of shard 0 (the packed per-rank slices of the all-gather, or plain [S][B][...] arrays).
struct MergeParams {
    const unsigned char *gain;     // shard 0: [B][k] fp64
    const unsigned char *row4col;  // shard 0: [B][k][ldCol] i32
    const unsigned char *nf;       // shard 0: [B] i32
    long long shardStride;         // bytes between consecutive shards' gain tables (and, unless the next two are set, of all three)
    long long strideR4C, strideNf; // 0 = shardStride (packed slices); else the tables' own strides (plain [S][B][...] arrays)
    int nShard, k, maxCol, ldCol, maximize;
    double *outGain;               // [B][k]
    int *outRow4col;               // [B][k][ldCol]
    int *outNf;                    // [B]
    int *outCol4row;               // [B][k][ldRow] or nullptr: the inverse of the merged row4col, -1 for rows without a column
    int ldRow;
    int inI8;                      // the shards' row4col tables are int8 (the merged table is int32 either way)
    int spd;                       // > 1: every block (shardStride apart) holds this many shards, tables of B problems back to back
};
// The merge from the shards' gains alone (kbest_merge.hip, merge_gains_kernel): all shards' gains and counts, the device's own
// shards' rows; out: merged gains and counts (whole), the own winners' rows at their merged positions in a byte table (zero
// elsewhere: a sum all-reduce completes it), and one word that is set when two candidates of a matrix have exactly the same gain.
struct MergeGainsParams {
    const unsigned char *gain;     // block 0: [spd][B][k] fp64; block j (a device's share of the all-gather) blockStride bytes behind
    const unsigned char *nf;       // block 0: [spd][B] i32
    long long blockStride;
    int spd;                       // shards per block (<= 1: one)
    const signed char *ownRow8;    // [ownHi - ownLo][B][k][maxCol]: the rows of shards ownLo .. ownHi-1
    int ownLo, ownHi;
    int nShard, B, k, maxCol, maximize;
    double *outGain;               // [B][k]
    signed char *outRow8;          // [B][k][maxCol], zeroed by the caller
    int *outNf;                    // [B]
    int *tied;                     // one word, zeroed by the caller
};
// kbest_batch_f64 with the tables staged in (and left in) caller-owned device buffers: the multi-device entry's per-device step
// (kbest_capi.cpp).  stamps (optional): host times (seconds, steady clock) at which the first upload / the first kernel was issued.
// row4col8 (optional): device memory for the block's row4col as bytes -- the narrow staging of uniform square batches then
// runs with it (bytes cross PCIe, col4row is rebuilt on the host and not kept).  keepI8 = 0: the table that is KEPT is the int32
// `row4col` (the narrow-staged path widens the bytes into it on the device); keepI8 = 1: the kept table is the byte table
// `row4col8` (a device's slice of the multi-device exchange, which travels in bytes) and `row4col` is int32 staging for the paths
// whose kernels write int32 (they narrow it into row4col8 on the device).
struct KeepTables { int32_t *row4col, *col4row; double *gain; int32_t *nf; double *stamps; signed char *row4col8; int keepI8; };
hipError_t launch_widen_i8(const signed char *src, int *dst, long long n, hipStream_t stream);
hipError_t launch_copy_words(const void *src, void *dst, long long bytes, hipStream_t stream);  // bytes: a multiple of 4
double now_s();
// kbest_exact.hip: kBest2D / kBest2DCutoff in the reference's own order of operations (padded N x N formulation, one heap of fully
// solved hypotheses with libstdc++'s sift rules), any size: KBEST_FLAG_REFERENCE_ORDER, and every problem beyond KBEST_MAX_DIM_WIDE rows
struct ExactParams {
    const double *cost;
    const long long *costOff;
    const int *nRow, *nCol;
    int B, maxRow, maxCol, ldRow, ldCol, k, maximize, useCutoff;
    unsigned flags;            // KBEST_FLAG_TABLES_I8
    double cutoff;
    int *row4col, *col4row;    // [B][k][ldCol] / [B][k][ldRow] (col4row may be null; padded columns named as the reference names them)
    double *gain;
    int *nf;                   // -4: the work space's pool of hypotheses ran out
    long long *pushed;         // or null
    unsigned char *work;       // gridDim.x slots of exact_slot_bytes(maxRow, hypPerSlot)
    int hypPerSlot;
    int childWaves;            // (set by launch_kbest_exact: waves of a workgroup that solve children, 65 .. 1 024 rows)
};
long long exact_slot_bytes(int maxRow, int hypPerSlot);
hipError_t launch_kbest_exact(const ExactParams &p, int grid, hipStream_t stream);
hipError_t launch_merge_topk(const MergeParams &p, int B, hipStream_t stream);
hipError_t launch_merge_gains(const MergeGainsParams &p, hipStream_t stream);
hipError_t launch_zero_words(unsigned *w, long long n, hipStream_t stream);
hipError_t launch_narrow_i32(const int *src, signed char *dst, long long n, hipStream_t stream);
hipError_t launch_add_i8(signed char *dst, const signed char *src, long long n, hipStream_t stream);
hipError_t launch_fill_unused(const int *nf, const int *nRow, const int *nCol, int B, int k, int ldCol, int ldRow, int *row4col,
                              int *col4row, double *gain, bool tablesI8, hipStream_t stream);
// The kernel behind every enumeration launch (kbest_merge.hip, kbest_ties.h): runs of equal gains into the canonical order,
// KBEST_TIE_* flags from tieGain (the gain of the solution behind the tables; nullptr: none was enumerated) into tieFlags (or
// nullptr), and -- fill -- the unused slots / the padding of a ragged batch as launch_fill_unused defines them.
hipError_t launch_finish_tables(const int *nf, const int *nRow, const int *nCol, int B, int k, int ldCol, int ldRow, int *row4col,
                                int *col4row, double *gain, bool tablesI8, const double *tieGain, int *tieFlags, bool fill, hipStream_t stream,
                                int baseFlags = 0, bool order = true);  // baseFlags: or-ed into every problem's flags (KBEST_TIE_UNCHECKED);
                                                                        // order = false: runs of equal gains are reported, not ordered (tables in host memory)
hipError_t launch_kbest_lane(const Params &p, int B, int nWaves, int lanesPerChild, hipStream_t stream);
hipError_t launch_kbest_small(const SmallParams &p, int B, int nWaves, hipStream_t stream);
// kbest_tiny.hip: the fused association path by exhaustive enumeration, for frames whose assignments are few (condition + gate +
// cutoff mode of SmallParams only); nf = -2: the frame is for the enumeration kernels after all
hipError_t launch_kbest_tiny(const SmallParams &p, int B, bool many, hipStream_t stream);
// kbest_bnb.hip: the fused association path by a bounded depth-first walk over the columns (every assignment whose partial sums
// stay below a bound that is raised until k assignments lie below it); same modes and -2 convention as kbest_tiny.hip
hipError_t launch_kbest_bnb(const SmallParams &p, int B, bool many, hipStream_t stream);
int bnb_lds_bytes(int k, int nThreads = 1024, int maxRow = 64, int maxCol = 16);
int tiny_lds_bytes(int k, int nThreads = 1024);
hipError_t launch_kbest(const Params &p, int B, int nWaves, hipStream_t stream);
hipError_t launch_kbest_wide(const WideParams &p, int grid, hipStream_t stream);
hipError_t launch_quadric_costs(const QuadricParams &p, int B, hipStream_t stream);
hipError_t launch_bb_costs(const BoxParams &p, int B, hipStream_t stream);
hipError_t launch_bb_assign(const BoxParams &p, const int *row4col, const int *nf, int k, int maxCol, int B,
                            hipStream_t stream);
hipError_t launch_weights(const WeightParams &p, int B, hipStream_t stream);
hipError_t launch_condition(const CondParams &p, int B, hipStream_t stream);
hipError_t launch_to_probs(double *x, long long n, hipStream_t stream);

}  // namespace kb

// kbest_capi.cpp: completes the gain levels that straddle slot k in the caller's HOST tables (see there)
#ifdef __cplusplus
#include <vector>
void kb_complete_tie_levels(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow, const int32_t *nCol,
                            const double *cost, const int64_t *costOff, int k, void *row4col, void *col4row, double *gain, int32_t *fl,
                            std::vector<int> *changed);
#endif
int kbest_batch_f64_keep(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow,
                         const int32_t *nCol, const double *cost, const int64_t *costOff, int k, int32_t *row4col,
                         int32_t *col4row, double *gain, int32_t *nf, int64_t *pushed, const kb::KeepTables *keep);
#endif
