/*
 * kbest_c.h -- C ABI of the MI355X-native k-best assignment engine.
 *
 * This is the drop-in boundary for the reference's k-best / association-weight
 * hot path.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository root):
 *
 *   kbest_batch_f64 / kbest_batch_f64_dev
 *        batched form of  kBest2D        (shortestPathCPP.hpp:204-212, cpp:571-644)
 *        and of           kBest2DCutoff  (shortestPathCPP.hpp:256-265, cpp:646-733)
 *        (opts.use_cutoff selects which); with k == 1 it is also the batched
 *        form of          assign2D       (shortestPathCPP.hpp:144-149) as used by
 *        asgnBB (assignment.cpp:750, k=1, maximize).
 *   kbest_weights_batch_f64
 *        batched form of  assignmentProb (assignment.h:11, assignment.cpp:547-683)
 *   kbest_condition_costs_f64
 *        batched form of  conditionCosts (assignment.h:26, assignment.cpp:439-525)
 *   kbest_assoc_probs_batch_f64
 *        batched form of  getAssignmentProbs from the cost matrix on (assignment.cpp:57-74)
 *   kbest_quadric_costs_f64 / kbest_quadric_assoc_probs_batch_f64
 *        batched form of  computeQuadricCostMatrix (assignment.cpp:705-722) and of the whole
 *        getAssignmentProbs chain behind it
 *   kbest_bb_match_batch_f64
 *        batched form of  asgnBB (assignment.cpp:724-797; k = 1, maximize)
 *
 * Conventions kept from the reference: cost matrices are column-major
 * C[row + col*numRow] with numRow >= numCol (shortestPathCPP.hpp:185-190);
 * row4col is indexed by column, col4row by row; col4row values >= numCol mean
 * "row sits on a zero-padded column" (SURVEY 8(a) quirk 6); the number of
 * solutions found is returned per problem, 0 = infeasible (cpp:588-593).
 * The solver never throws; negative return values are engine errors.
 *
 * Order of exact ties.  Hypotheses with EXACTLY equal gain have no defined relative order in the reference (it is an
 * artefact of std::priority_queue's binary heap, shortestPathCPP.cpp:30-42, 574; which of them fill the last slots of a
 * call is an artefact too).  For continuous costs ties have probability zero and every output is the reference's, bit for bit.
 * For integer-like costs (conditionCosts produces exact zeros) there are two answers to choose from:
 *   (1) THE REFERENCE'S OWN -- what its heap pops, slot for slot.  The SYNCHRONOUS k-best entries (kbest_batch_f64,
 *       kbest_resolve_ties_dev behind the asynchronous entry, kbest_batch_f64_multi in batch mode) give it BY DEFAULT: the batch runs
 *       on the fast kernels, every enumeration launch enumerates ONE solution more than asked for (its gain only; measured free) and
 *       is followed by a small launch that reports, per problem, KBEST_TIE_* flags (kbest_opts.tie_flags): every problem with two
 *       exactly equal gains among its k + 1 best -- inside the table or across slot k -- is then enumerated AGAIN by the
 *       reference-order kernel (kbest_exact.hip: the reference's algorithm as it stands) and its tables are replaced
 *       (KBEST_TIE_REFERENCE).  A tie-free problem has ONE sequence of k best, which the fast kernels return bit for bit, so the call
 *       as a whole answers as the reference does, at the fast kernels' speed wherever nothing ties.  KBEST_FLAG_REFERENCE_ORDER runs
 *       EVERY problem on that kernel (slow, exact; also names col4row's padded columns as the reference does on every problem).
 *   (2) THE ENGINE'S RULE, the same in every kernel and for every batch a problem may travel in: solutions ordered by
 *       (gain, row4col), row4col compared lexicographically in the reference's column order; when the k-th and the (k+1)-th best
 *       gains are equal -- the k best are then not a unique set -- the lexicographically first assignments of that gain level are
 *       kept.  The multiset of gains and the validity of every assignment are the reference's; the order inside a run of equal gains
 *       and the members of a level that straddles slot k are the rule's.  This is what the finishing launch leaves in the tables, so
 *       what the ASYNCHRONOUS entries return (with the flags), what the association entries weigh by default (the exhaustive kernel
 *       and the bounded walk see a whole gain level and keep its lexicographically first members themselves, whatever its size), and
 *       what the synchronous entries return with KBEST_FLAG_CANONICAL_TIES: a tie at slot k (KBEST_TIE_BOUNDARY) is then completed by
 *       enumerating the problem again with k + 64, then k + 256, k + 1 024, k + KBEST_TIE_CAP solutions until the level ends inside
 *       the table (KBEST_TIE_RESOLVED; a level with more than KBEST_TIE_CAP members beyond k -- on the association entries, whose
 *       weights are summed on the device: of more than 4 096 members in all -- stays KBEST_TIE_UNRESOLVED: the emitted set is then one
 *       of several equally good ones and may depend on the kernel; a re-run that fails leaves the first pass' tables and that flag).
 *       On tie-heavy batches (1) is also the cheaper one: one more run of k solutions instead of thousands of members of a level.
 * KBEST_FLAG_NO_TIE_RESOLVE: the synchronous entries only report the flags; KBEST_FLAG_NO_TIE_CHECK switches all of it off (the
 * kernels' own orders, round 4).  Association entries: kbest_set_reference_order(ctx, 2) weighs the reference's own k best on frames
 * with a tie at slot k (1: on every frame).
 * Index outputs are int32 (the reference ABI uses ptrdiff_t; the C++ shims in
 * include/kbest_shims.hpp widen on the host).
 *
 * All compute runs in hand-written HIP kernels for gfx950
 * (probabilisticsemslam_amd/csrc/kbest_engine.hip).  There is no CPU fallback:
 * without a GPU every compute entry point returns KBEST_ERR_NO_DEVICE.
 */
#ifndef KBEST_C_H
#define KBEST_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kbest_ctx kbest_ctx; /* opaque: device, stream, workspace pools */

enum {
    KBEST_OK = 0,
    KBEST_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime error at create    */
    KBEST_ERR_BAD_ARG = -2,     /* null pointer, k < 1, numRow < numCol, ...      */
    KBEST_ERR_UNSUPPORTED = -3, /* numRow > KBEST_MAX_DIM_EXACT                                                     */
    KBEST_ERR_HIP = -4,         /* a HIP call failed; see kbest_last_error()      */
    KBEST_ERR_NOMEM = -5,
    KBEST_ERR_NOT_RESERVED = -6, /* kbest_batch_f64_dev: workspace too small, call kbest_reserve first   */
    KBEST_ERR_INTERNAL = -7     /* a problem came back with nf < 0 although its shape was accepted       */
};

#define KBEST_MAX_DIM 64       /* rows per problem handled by the LDS-resident kernel (the fast path)      */
#define KBEST_MAX_DIM_WIDE 1024 /* rows per problem of the general-size kernel (beyond KBEST_MAX_DIM, or a k beyond the LDS    */
                               /* candidate pool: HBM work space)                                                         */
#define KBEST_MAX_DIM_EXACT 16384 /* rows per problem handled at all: beyond KBEST_MAX_DIM_WIDE the reference-order kernel runs */
                               /* (kbest_exact.hip: the reference's algorithm as it stands, one wave per problem -- slow, total; */
                               /* its work space holds one record of 25 numRow bytes per pushed hypothesis: KBEST_ERR_NOMEM    */
                               /* where 1 + (k - 1) numCol of them do not fit 16 GiB)                                         */

/* flags */
#define KBEST_FLAG_NO_PRUNE 1u     /* disable early termination (for counting P)   */
#define KBEST_FLAG_COUNT_PUSHED 2u /* fill `pushed` with the reference's push count */
#define KBEST_FLAG_RECT_ROOT 4u    /* internal (kbest_assign_batch_f64): numCol augmentations on the rectangular problem */
#define KBEST_FLAG_NO_SHIFT 8u     /* internal (kbest_assign_batch_f64): the cost matrix is already non-negative         */
#define KBEST_FLAG_NO_T0 32u       /* do not use the a-priori threshold from combinations of the root's children (A/B tests) */
#define KBEST_FLAG_EXACT_ROOT 16u  /* root LAP by the reference's own sequence of augmentations (no column reduction first)  */
#define KBEST_FLAG_NO_REORDER 128u /* 64-row kernel: enumerate in the reference's column order (A/B tests; same results)          */
#define KBEST_FLAG_NO_OPT 256u     /* 64-row kernel: no optimistic bounds / re-split tickets (A/B tests; same results)           */
#define KBEST_FLAG_NO_TIE_CHECK 512u /* do not enumerate the (k+1)-th solution / order exact ties canonically (see "Order of exact ties") */
#define KBEST_FLAG_REFERENCE_ORDER 2048u /* kbest_batch_f64[_dev]: the REFERENCE's own order of operations (kbest_exact.hip) -- the   */
                                        /* zero-padded N x N formulation, one priority queue of fully solved hypotheses with libstdc++'s */
                                        /* sift rules: hypotheses with exactly equal gains come out in the order the reference's heap  */
                                        /* pops them (shortestPathCPP.cpp:30-42, 574) and col4row names the padded column of every      */
                                        /* left-over row as the reference does.  Slow (up to 8 waves per problem, nothing pruned): for */
                                        /* callers with integer-like costs who need the reference's answer slot for slot.  No tie flags. */
#define KBEST_FLAG_REFERENCE_TIES 4096u /* (accepted; the DEFAULT of the synchronous k-best entries since round 6: every problem whose    */
                                       /* k + 1 best gains hold an exact tie is enumerated again by the reference-order kernel and its  */
                                       /* tables replaced, KBEST_TIE_REFERENCE: "Order of exact ties" (1))                              */
#define KBEST_FLAG_CANONICAL_TIES 8192u /* synchronous k-best entries, kbest_resolve_ties_dev, the multi-device batch entry: the        */
                                       /* engine's own rule on exact ties instead of the reference's answer: "Order of exact ties" (2)  */
#define KBEST_FLAG_NO_TIE_RESOLVE 1024u /* synchronous entries: report a tie at slot k (KBEST_TIE_BOUNDARY), do not complete its gain level */
#define KBEST_FLAG_TABLES_I8 64u   /* kbest_batch_f64 / kbest_batch_f64_dev: row4col / col4row are tables of int8_t (same shapes, */
                                   /* same values, -1 = unassigned / unused) instead of int32_t: every index of a problem of up   */
                                   /* to 127 rows fits a byte, and a quarter of the bytes cross PCIe.  numRow > 127:              */
                                   /* KBEST_ERR_UNSUPPORTED.  Not with the multi-GPU entries: the caller's tables are int32 there  */
                                   /* (what travels BETWEEN the devices is in bytes by itself wherever the indices fit them).      */

/* per-problem tie flags (kbest_opts.tie_flags, kbest_set_assoc_tie_flags_dev, kbest_last_tie_flags; see "Order of exact ties" above) */
#define KBEST_TIE_INSIDE 1            /* some of the emitted gains are exactly equal (they are in the canonical order)        */
#define KBEST_TIE_BOUNDARY 2          /* the k-th and the (k+1)-th best gains are exactly equal: the k best are not unique    */
#define KBEST_TIE_RESOLVED 4          /* ... and the entry completed that gain level: the lexicographically first were kept   */
#define KBEST_TIE_REFERENCE 8         /* KBEST_FLAG_REFERENCE_TIES: the problem was enumerated again by the reference-order kernel: its tables are the reference's own */
#define KBEST_TIE_UNCHECKED (1 << 28) /* ASYNCHRONOUS entries only: k sits at the limit of the kernel the problem ran on (e.g.     */
                                      /* k = 1 024 on the fused association kernel): the (k+1)-th solution was not enumerated, so */
                                      /* a tie at slot k would not have been seen (runs of equal gains INSIDE the tables are       */
                                      /* ordered as ever).  The synchronous entries take a kernel that fits k + 1 instead.         */
#define KBEST_TIE_UNORDERED (1 << 29) /* a run of more than 4 096 equal gains was left in the kernel's own order (asynchronous   */
                                      /* and association entries; kbest_batch_f64 orders such a run on the host)                */
#define KBEST_TIE_UNRESOLVED (1 << 30) /* BOUNDARY without RESOLVED: the emitted set is one of several equally good ones       */
#define KBEST_TIE_CAP 4096            /* members of the gain level at slot k beyond k that a synchronous entry enumerates at     */
                                      /* most (it tries 64, 256, 1 024, 4 096)                                                  */

/* ALWAYS initialise a kbest_opts with kbest_default_opts() and then set what you need: the struct has grown at its end
 * (tie_flags, round 5) and may grow again; a struct filled member by member by code compiled against an older header -- or one
 * that was never initialised -- hands the engine an indeterminate tie_flags pointer, which the finishing launch WRITES through. */
typedef struct kbest_opts {
    int32_t  maximize;     /* reference `maximize` argument                       */
    int32_t  use_cutoff;   /* 0: kBest2D semantics; 1: kBest2DCutoff semantics    */
    double   cutoff;       /* reference `cutoff` argument (assignment.cpp:9: 42)  */
    uint32_t flags;        /* KBEST_FLAG_*                                        */
    int32_t  root_col_offset; /* subtree sharding (multi-GPU latency mode):       */
    int32_t  root_col_stride; /* only root children on columns c with             */
                              /* c % stride == offset are expanded; 0/1 = all.    */
                              /* c is the REFERENCE's column and the partition is the reference's (split, cpp:455-532: the    */
                              /* child on column c keeps the root's rows on columns 0 .. c-1): a sharded run enumerates in    */
                              /* the reference's column order whatever kernel or launch shape runs it, so shards computed by  */
                              /* differently configured ranks still partition the problem.                                     */
    int32_t *tie_flags;       /* [B] KBEST_TIE_* per problem, or NULL: a host pointer for the host-buffer entries, a device     */
                              /* pointer for the _dev entries                                                                  */
} kbest_opts;

void kbest_default_opts(kbest_opts *o);

int kbest_create(kbest_ctx **ctx, int device);
int kbest_destroy(kbest_ctx *ctx);
const char *kbest_strerror(int code);
const char *kbest_last_error(const kbest_ctx *ctx);
int kbest_device_count(void);

/*
 * Batched k-best, buffers already resident in device memory (HBM).
 *   B         number of problems
 *   maxRow/maxCol  upper bounds of the shapes in the batch; also the leading
 *             dimensions of the outputs
 *   d_nRow/d_nCol  per-problem shapes, or NULL for uniform maxRow x maxCol
 *   d_cost    packed column-major cost blocks; problem b starts at
 *             d_costOff[b] doubles (or b*maxRow*maxCol when d_costOff is NULL)
 *   d_row4col [B][k][maxCol] int32   col -> row   (row4colBest, hpp:228-229)
 *   d_col4row [B][k][maxRow] int32   row -> col   (col4rowBest, hpp:226-227); NULL = not wanted
 *   d_gain    [B][k] double                       (gainBest,    hpp:230-231)
 *   d_nf      [B] int32   number found, 0 = infeasible (return value of kBest2D)
 *   d_pushed  [B] int64 or NULL; with KBEST_FLAG_COUNT_PUSHED the number of
 *             feasible children the reference would push (SURVEY 8(d) "P")
 *   opts->tie_flags  [B] int32 in device memory or NULL: KBEST_TIE_* per problem (this entry reports a tie at slot k,
 *             it cannot complete it: no synchronisation inside)
 *   stream    hipStream_t (NULL = the context's own stream).  Asynchronous:
 *             returns after enqueueing; no host synchronisation and no allocation
 *             inside -- the workspace must have been sized by kbest_reserve(B, maxRow, k)
 *             beforehand (else KBEST_ERR_NOT_RESERVED), which keeps the entry legal
 *             inside a graph capture.
 * ONE hypothesis workspace per context: launches of one context execute one after the
 * other.  A launch on a different stream than the previous one is ordered behind it
 * with an event (so results stay correct), but there is no overlap to be had -- use one
 * context per stream (or per host thread) for concurrent launches.
 */
int kbest_batch_f64_dev(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                        const int32_t *d_nRow, const int32_t *d_nCol, const double *d_cost,
                        const int64_t *d_costOff, int k, int32_t *d_row4col, int32_t *d_col4row,
                        double *d_gain, int32_t *d_nf, int64_t *d_pushed, void *stream);
/*
 * The second call for callers of kbest_batch_f64_dev whose costs can tie exactly (integer-like costs): SYNCHRONOUS.  Same
 * arguments as the launch it follows (same opts, shapes, cost blocks, tables -- all still on the device), d_tie_flags = the
 * opts->tie_flags the launch wrote.  Waits for `stream` and does what the synchronous entries do with flagged problems ("Order of exact
 * ties"): by default every problem flagged with a tie (inside its table, across slot k, or unchecked) is enumerated again by the
 * reference-order kernel and its slots of d_row4col / d_col4row / d_gain are replaced (KBEST_TIE_REFERENCE: the reference's own answer);
 * with KBEST_FLAG_CANONICAL_TIES every problem flagged KBEST_TIE_BOUNDARY has its gain level at slot k completed under the engine's rule
 * (the problem again with k + 64 / 256 / 1 024 / KBEST_TIE_CAP solutions; the first k of the canonically ordered table written over the
 * problem's slots; KBEST_TIE_RESOLVED, or KBEST_TIE_UNRESOLVED where the level is larger than the cap).  A batch without flagged
 * problems costs one small copy.
 */
int kbest_resolve_ties_dev(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *d_nRow,
                           const int32_t *d_nCol, const double *d_cost, const int64_t *d_costOff, int k, int32_t *d_row4col,
                           int32_t *d_col4row, double *d_gain, int32_t *d_tie_flags, void *stream);

/*
 * Same with host buffers (copies in, runs, copies out, synchronises; col4row may be NULL = not wanted).  Uniform square batches
 * of up to KBEST_MAX_DIM rows go through narrow staging: the kernels write row4col as bytes into pinned memory of the context, in
 * pieces, and host threads of the context widen a piece into the caller's int32 row4col and its inverse col4row while the GPU
 * works on the next one (the same tables bit for bit; 1 024 x 64x64, k = 200, 107 MB of int32 tables: 2.5 - 2.8 ms per call, the
 * kernel alone 1.8; round 3's paths -- copying 4.3, kernels writing the int32 tables into registered memory 3.1 -- remain for
 * everything else and behind KBEST_NO_NARROW).  A caller that runs batch after batch with the same buffers can register them
 * once (kbest_register_host_buffer): result tables that lie in registered memory are written there by the kernels themselves,
 * spread over the whole run; cost blocks in registered memory are read in place (each block once, into its LDS tile).  With
 * KBEST_FLAG_TABLES_I8 the caller takes the byte tables as they are (2.2 ms).
 */
int kbest_batch_f64(kbest_ctx *ctx, const kbest_opts *opts, int B, int maxRow, int maxCol,
                    const int32_t *nRow, const int32_t *nCol, const double *cost,
                    const int64_t *costOff, int k, int32_t *row4col, int32_t *col4row, double *gain,
                    int32_t *nf, int64_t *pushed);

/*
 * Batched assign2D (shortestPathCPP.hpp:144-149, cpp:735-762) and shortestPathCPP (hpp:178-182, cpp:119-238): the
 * single best assignment of each numRow x numCol problem by numCol shortest-augmenting-path steps on the
 * RECTANGULAR matrix (no zero-padded columns), with the dual variables.
 *   shift      1: assign2D -- the matrix is made non-negative first (makeCostMatrixSafe) and the gain is un-shifted;
 *              0: shortestPathCPP -- the matrix is used as it is (it must be non-negative; maximize must be 0)
 *   gainCols   numCol4Gain of shortestPathCPP (the gain sums the first gainCols columns, cpp:232); 0 = numCol
 *   row4col    [B][maxCol]  col -> row          col4row [B][maxRow]  row -> col, -1 = unassigned (cpp:134)
 *   u          [B][maxCol] MurtyHyp::u (per column), v [B][maxRow] MurtyHyp::v (per row); either may be NULL
 *   feasible   [B] 1, or 0 = infeasible (then gain = -1, cpp:197-203, and the index outputs are -1)
 * numRow <= KBEST_MAX_DIM.  Host buffers.
 */
int kbest_assign_batch_f64(kbest_ctx *ctx, int B, int maxRow, int maxCol, const int32_t *nRow, const int32_t *nCol,
                           const double *cost, const int64_t *costOff, int maximize, int shift, int gainCols,
                           int32_t *row4col, int32_t *col4row, double *gain, double *u, double *v,
                           int32_t *feasible);

/* toProbs (assignment.h:19, assignment.cpp:527-542): x[i] <- exp(min(x) - x[i]) where min(x) + 42 > x[i], else 0;
 * in place on a host buffer of n doubles. */
int kbest_to_probs_f64(kbest_ctx *ctx, double *x, int64_t n);

/* Size the context's workspace for launches of up to B problems of up to maxRow rows and k solutions.
 * Required before kbest_batch_f64_dev; the host-pointer entries call it implicitly.  When it has to
 * grow the workspace it waits for the device to go idle (hipDeviceSynchronize) and reallocates.
 * The workspace is: the hypothesis states ([B][k + 64 + up to 1 024] records of 18 maxRow + 24 bytes), the slot tables, the
 * gains behind the tables (exact ties) and -- for batches of more than one generation of resident workgroups, which the 64-row
 * kernel enumerates as a relay of several workgroups per matrix -- one LDS image per matrix (25 KB at 32 rows, 70 KB at 64;
 * at most ~400 MB: larger batches are not relayed).  A kbest_batch_f64_dev call whose batch was not reserved for fails with
 * KBEST_ERR_NOT_RESERVED; one whose relay images alone are missing (reserved with a smaller B) runs as a plain launch.
 * A graph that captured a relay launch holds the addresses of the relay work space: reserve for the largest batch BEFORE
 * capturing -- a later kbest_reserve that would have to grow that work space returns KBEST_ERR_BAD_ARG.  A piece of a relay
 * waits for its predecessor at most a few seconds (a healthy launch never gets there); a matrix whose hand-over did not come is
 * reported with nf = -3 (KBEST_ERR_INTERNAL from the host entries), and the context re-zeroes its progress words before the
 * next relay launch after any failed entry. */
int kbest_reserve(kbest_ctx *ctx, int B, int maxRow, int k);
/* The same for launches that run on the reference-order kernel (KBEST_FLAG_REFERENCE_ORDER, or maxRow > KBEST_MAX_DIM_WIDE): its work
 * space -- per resident problem a padded cost copy (8 maxRow^2 bytes) and a pool of 2 + (k - 1) maxCol hypotheses of 25 maxRow bytes.
 * kbest_batch_f64_dev needs it up front for such launches; the host entries grow it on demand. */
int kbest_reserve_exact(kbest_ctx *ctx, int B, int maxRow, int maxCol, int k);

/* Diagnostic builds only (make -C probabilisticsemslam_amd/csrc PROFILE=1): device buffer of B*16 uint64
 * that receives per-matrix cycle stamps of the kernel phases.  The regular build never touches it. */
int kbest_set_profile_buffer(kbest_ctx *ctx, void *d_buf);

/*
 * Batched assignmentProb (assignment.cpp:547-683): k-best with cutoff 42, then
 * sum of exp(best - cost) over the solutions scattered into probs.
 *   nL[b], nM[b]   landmarks / measurements of problem b; its cost block is
 *                  (nL+nM) x nM column-major
 *   probs          packed, problem b at probOff[b] doubles: [nM][nL+1] row-major
 *                  (the reference's vector<vector<double>>)
 * Host buffers.
 */
int kbest_weights_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM,
                            const double *cost, const int64_t *costOff, int k, double *probs,
                            const int64_t *probOff, int32_t *nf);

/*
 * The accumulation of bruteForceProb (assignment.h:43, assignment.cpp:880-945): plain kBest2D with the caller's k
 * (the reference derives it from a Minc-type bound, :858-868 -- the shim in kbest_shims.hpp does the same) and the
 * weights summed WITHOUT the best+42 mask.  Same layout as kbest_weights_batch_f64.
 */
int kbest_bruteforce_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM,
                                     const double *cost, const int64_t *costOff, int k, double *probs,
                                     const int64_t *probOff, int32_t *nf);

/*
 * Batched conditionCosts (assignment.h:26, assignment.cpp:439-525).  Problem b: nRow[b] x nCol[b] column-major
 * at cost + costOff[b].  out receives the conditioned goodRows[b] x nCol[b] block at the same offset; rowIdx
 * [B][maxRow] the original row of each kept row (rowIdxOut of the reference; unused tail = -1).  Host buffers.
 */
int kbest_condition_costs_f64(kbest_ctx *ctx, int B, const int32_t *nRow, const int32_t *nCol,
                              const double *cost, const int64_t *costOff, double *out, int32_t *goodRows,
                              int32_t *rowIdx, int maxRow);

/*
 * Cost block in, association probabilities out: conditionCosts -> assignmentProb(k) -> scatter back to the
 * original landmark numbering, i.e. getAssignmentProbs (assignment.cpp:57-74) from the cost matrix on, all on
 * the device.  Same argument layout as kbest_weights_batch_f64; cost blocks are the UNconditioned
 * (nL+nM) x nM matrices of computeQuadricCostMatrix (assignment.cpp:705-722); probs is [nM][nL+1] per problem.
 * One launch for frame-sized blocks.  Frames of up to 16 measurements and 64 rows (all frames of the call) are not
 * enumerated at all: conditioned costs are >= 0, so a walk over the columns that drops every partial assignment whose
 * partial sum exceeds a bound visits exactly the assignments below the bound; the bound is raised until k assignments
 * lie below it and the k cheapest of those are the answer (kbest_bnb.hip) -- gains bit for bit calcGain's sums, same
 * solutions, same probabilities as the enumeration; KBEST_NO_BNB=1 at kbest_create switches it off.  Larger frames take
 * the fused association kernel (kbest_small.hip), and -- when every frame of the
 * call has so few assignments in all, (nL+nM)!/nL! <= 2^23 (and <= 2^15 choices for the first nM-2 columns) with
 * 2 <= nM <= 8 and nL+nM <= 64: the reference's real
 * frames of 3-5 measurements (README.md:11) -- the exhaustive kernel (kbest_tiny.hip), which looks at every assignment
 * instead of enumerating the k best: same gains bit for bit (calcGain's sum), same solutions, same probabilities -- exact
 * ties included ("Order of exact ties" above: the flags of these entries are read with kbest_last_tie_flags).  KBEST_NO_TINY=1 at kbest_create
 * switches it off.  kbest_weights_batch_f64 on blocks that are conditioned already (all entries >= 0, an exact zero
 * somewhere: what conditionCosts returns and assignment.cpp:58-62 passes on) takes the same kernel; any other block
 * is answered by the enumeration kernels.
 */
int kbest_assoc_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM,
                                const double *cost, const int64_t *costOff, int k, double *probs,
                                const int64_t *probOff, int32_t *nf);

/*
 * The same on device buffers, asynchronous on `stream` (NULL = the context's stream): ONE launch -- of the bounded walk
 * (kbest_bnb.hip) when maxRawRow <= 64 and maxCol <= 16 (followed by a launch of the enumeration kernel that looks only at
 * what the walk handed back: frames with masses of equal gains), else of the fused association kernel (kbest_small.hip):
 * conditionCosts while the cost tile is loaded, the k best within the cutoff 42, the exp-weights, the scatter back --
 * for callers whose cost blocks are produced on the GPU.
 *   d_nRow[b] = d_nL[b] + d_nM[b] rows of the cost block of frame b; condition = 0: the blocks are already
 *   conditioned (assignmentProb only).  Limits of the fused kernel: nM <= 32, k <= 1024, at most 32 rows kept by
 *   conditionCosts (a frame beyond that comes back with d_nf[b] = -2 and zero probabilities: re-run it through the
 *   host-pointer entry, which falls back to the general pipeline by itself).  Needs kbest_reserve_assoc first.
 */
int kbest_assoc_probs_batch_f64_dev(kbest_ctx *ctx, int B, int maxRawRow, int maxCol, const int32_t *d_nL,
                                    const int32_t *d_nM, const int32_t *d_nRow, const double *d_cost,
                                    const int64_t *d_costOff, int k, int condition, double *d_probs,
                                    const int64_t *d_probOff, int32_t *d_nf, void *stream);
int kbest_reserve_assoc(kbest_ctx *ctx, int B, int maxRawRow, int maxCol, int k);
/* on = 1: the HOST-buffer association entries of this context (kbest_weights / assoc_probs / bruteforce / quadric_assoc) enumerate
 * their k best in the REFERENCE's own order of operations (the reference-order kernel, as KBEST_FLAG_REFERENCE_ORDER does for
 * kbest_batch_f64): where exactly equal gains straddle slot k the assignments that are weighed are the ones the reference's
 * kBest2DCutoff returns (assignment.cpp:594), so the probabilities are the reference's there too (integer-like costs).  Slower: no
 * fused kernels.  The reference-named shims switch it on with KBEST_SHIM_REFERENCE_ORDER=1.
 * on = 2: the same answer at the fused kernels' speed wherever nothing ties (as KBEST_FLAG_REFERENCE_TIES does for kbest_batch_f64):
 * the batch runs on the fused kernels, and only the frames whose k-th and (k+1)-th gains are exactly equal -- the only frames whose
 * weights depend on the order of ties: inside a run of equal gains every order sums the same terms -- are enumerated again by the
 * reference-order kernel and weighed from its table (KBEST_TIE_REFERENCE in kbest_last_tie_flags).  KBEST_SHIM_REFERENCE_ORDER=2.
 * on = 0: the engine's own rule (the default). */
int kbest_set_reference_order(kbest_ctx *ctx, int on);
/* Where kbest_assoc_probs_batch_f64_dev writes its frames' KBEST_TIE_* flags ([B] int32 in device memory; NULL, the default:
 * nowhere).  Stays set until changed. */
int kbest_set_assoc_tie_flags_dev(kbest_ctx *ctx, int32_t *d_flags);
/* KBEST_TIE_* flags of the problems of the context's last SYNCHRONOUS call (kbest_batch_f64 and every host-buffer association
 * entry, which have no other way to return them: the shims of kbest_shims.hpp included).  Copies min(n, cap) flags, returns n. */
int kbest_last_tie_flags(kbest_ctx *ctx, int32_t *flags, int cap);
/* Diagnostic: launches of the 64-row kernel this context has made as a relay (several workgroups per matrix in turn; the
 * comment of kbest_reserve, NOTES.md 10.6) since it was created -- for tests that must know the path they exercise was taken. */
long long kbest_relay_launches(kbest_ctx *ctx);  /* (-1: null context) */
/* Diagnostic: the kernel(s) the context's last k-best launch (kbest_batch_f64[_dev] and whatever goes through them) was routed
 * to -- KBEST_ROUTE_* bits -- for tests that must know the path they exercise was taken.  -1: null context. */
#define KBEST_ROUTE_LANE 1    /* the lane-per-child kernel: <= 32 rows, dense batches                   */
#define KBEST_ROUTE_SMALL 2   /* the small-problem kernel: <= 32 rows, rectangular / chip-underfilling  */
#define KBEST_ROUTE_FAST 4    /* the 64-row kernel                                                      */
#define KBEST_ROUTE_WIDE 8    /* the general-size kernel: any size, any k                               */
#define KBEST_ROUTE_RELAY 16  /* ... as a relay of several workgroups per matrix                        */
#define KBEST_ROUTE_EXACT 64  /* the reference-order kernel (KBEST_FLAG_REFERENCE_ORDER; > 1 024 rows)   */
#define KBEST_ROUTE_EXTRA 32  /* the launch enumerated the solution behind the k-th (exact ties checked) */
int kbest_last_route(kbest_ctx *ctx);

/*
 * Batched computeQuadricCostMatrix (assignment.h:28-29, assignment.cpp:705-722).  Frame b has nL[b] landmarks and
 * nM[b] measurements, each a (mean[3], cov[3][3] row-major) pair, packed frame after frame; gate is
 * NONASSIGN_QUADRIC.  cost receives the (nL+nM) x nM column-major blocks packed back to back.  Host buffers.
 */
int kbest_quadric_costs_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM, const double *landMean,
                            const double *landCov, const double *measMean, const double *measCov, double gate,
                            double *cost);

/*
 * getAssignmentProbs from the (mean, covariance) pairs on (assignment.cpp:42-74 after getMeans/getCovs): cost
 * construction, conditionCosts, assignmentProb(k), scatter back -- one stream-ordered device pipeline, the cost
 * matrix never visits the host.  probs: [nM][nL+1] per frame at probOff[b].
 */
int kbest_quadric_assoc_probs_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nM,
                                        const double *landMean, const double *landCov, const double *measMean,
                                        const double *measCov, double gate, int k, double *probs,
                                        const int64_t *probOff, int32_t *nf);

/*
 * Batched asgnBB (assignment.h:21, assignment.cpp:724-797): stereo bounding-box matching.  Boxes are
 * (xmin, ymin, xmax, ymax, xOffset) -- boundBox.h:13-25 -- nL[b] left and nR[b] right boxes per frame, packed;
 * gate is NONASSIGN_BOUNDBOX.  assign[sum nL]: for every left box the index of its right box, or -1.
 */
int kbest_bb_match_batch_f64(kbest_ctx *ctx, int B, const int32_t *nL, const int32_t *nR, const double *boxL,
                             const double *boxR, double gate, int32_t *assign);

/*
 * Pin [ptr, ptr + bytes) of caller-owned host memory and map it into the device's address space (hipHostRegister), for the
 * host-buffer entries above.  The range must stay allocated until kbest_unregister_host_buffer (or kbest_destroy).  Any number
 * of ranges; a buffer argument is taken as registered when it lies completely inside one of them.  Honoured by
 * kbest_batch_f64 (cost, row4col, col4row, gain, nf) and by the association entries (cost, probs).
 */
int kbest_register_host_buffer(kbest_ctx *ctx, void *ptr, size_t bytes);
int kbest_unregister_host_buffer(kbest_ctx *ctx, void *ptr);

/*
 * The "global k-best heap" of the subtree-sharded enumeration (reference partition: split,
 * shortestPathCPP.cpp:455-532): shard s of nShard enumerated the root's children on columns c % nShard == s
 * (kbest_opts.root_col_offset / root_col_stride) and holds its own k best -- slot 0 is the root on every shard.  This
 * k-way merge on the device gives the global table: the root, then the k - 1 best of the union of the shards' slots 1..
 * in increasing cost (decreasing profit when `maximize`); exact ties are ordered by the assignment (lexicographic
 * row4col), so the result does not depend on the number of shards.  d_gain / d_row4col / d_nf point at shard 0's
 * [B][k] fp64 / [B][k][maxCol] i32 / [B] i32 tables, shard s's tables start s * shardStrideBytes behind them (the
 * packed per-rank slices of the all-gather, or plain [nShard][B][...] arrays).  Asynchronous on `stream`
 * (NULL = the context's stream); device pointers.
 */
int kbest_merge_topk_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const void *d_gain,
                             const void *d_row4col, const void *d_nf, int64_t shardStrideBytes, double *d_outGain,
                             int32_t *d_outRow4col, int32_t *d_outNf, void *stream);
/* The same with the shards' row4col tables as int8 ([B][k][maxCol] bytes: what KBEST_FLAG_TABLES_I8 launches write and what the
 * exchange of problems of up to 127 rows moves); the merged table is int32. */
int kbest_merge_topk_i8_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const void *d_gain,
                                const void *d_row4col8, const void *d_nf, int64_t shardStrideBytes, double *d_outGain,
                                int32_t *d_outRow4col, int32_t *d_outNf, void *stream);
/*
 * The global k-best heap from the shards' COSTS alone (the north star's "allgather of per-rank top-k costs into a global k-best
 * heap"): d_gain [nShard][B][k] fp64 and d_nf [nShard][B] i32 are what an all-gather of every rank's (gain, nf) leaves on every
 * rank -- 8 k + 4 bytes per matrix and shard --; d_ownRow4col8 [B][k][maxCol] int8 are the rows of THIS rank's shard `ownShard`
 * (a KBEST_FLAG_TABLES_I8 launch with root_col_offset / stride).  Writes the merged gains d_outGain [B][k] and counts d_outNf [B]
 * in full (identical on every rank), and into d_outRow4col8 [B][k][maxCol] -- which the caller has ZEROED -- the rows of this
 * rank's own winners at their merged positions: ONE sum all-reduce of that byte table over the ranks (k maxCol bytes per matrix,
 * whatever the number of ranks; every entry is non-zero on at most one rank) completes it everywhere; slots beyond d_outNf stay 0.
 * *d_tied (one int32, zeroed by the caller) is set when two candidates of some matrix have EXACTLY the same gain within the k best
 * or at slot k: their order is the assignments' ("Order of exact ties" above), which this merge does not see -- the caller then
 * exchanges the whole lists and merges with kbest_merge_topk_i8_f64_dev.  numRow <= 127.  Asynchronous on `stream`.
 */
int kbest_merge_gains_f64_dev(kbest_ctx *ctx, int B, int nShard, int k, int maxCol, int maximize, const double *d_gain,
                              const int32_t *d_nf, int ownShard, const int8_t *d_ownRow4col8, double *d_outGain,
                              int8_t *d_outRow4col8, int32_t *d_outNf, int32_t *d_tied, void *stream);

/*
 * Multi-device entries (BASELINE.json config 4): one engine context + one stream per GPU and an
 * RCCL communicator over them (ncclCommInitAll; xGMI inside a node).  kbest_batch_f64_multi shards the batch in
 * contiguous blocks (device g solves matrices [g*ceil(B/G), ...)), each device solving its block with the kernels of
 * the single-device entries straight into its packed slice (gain | row4col | nf) of a global table; ONE in-place
 * ncclAllGather of those slices then leaves EVERY device with the same global k-best table (there is no
 * other collective: the matrices are independent).  Between the devices row4col travels as int8 wherever every index fits a
 * byte (numRow <= 127: 8 + numCol instead of 8 + 4 numCol bytes per solution); the caller's tables are int32 as ever.  Every device is fed and read back by a host thread of its own: the
 * host outputs of a block come from the device that solved it (col4row is not part of the exchange).  device_ids may
 * name one GPU several times ("logical devices", e.g. {0, 0, 0, 0}): the slices then travel by device-to-device copies
 * instead of RCCL -- the same host path, testable on one GPU.  Same argument meaning as kbest_batch_f64
 * (uniform packing b*maxRow*maxCol; nRow/nCol optional).  Exact ties: every device's tables come back in the one order of
 * equal gains ("Order of exact ties" above); a gain level that straddles slot k is NOT completed by these entries (their tables
 * stay on the devices for the exchange) and no flags are returned: a caller with integer-like costs that needs the canonical
 * members of such a level runs kbest_batch_f64 on the problems kbest_batch_f64_dev / kbest_batch_f64 flag.
 * RCCL is bound at run time (dlopen): without it
 * kbest_create_multi returns KBEST_ERR_NO_DEVICE and every single-device entry still works.
 */
typedef struct kbest_multi kbest_multi;
int kbest_create_multi(kbest_multi **m, const int *device_ids, int nDev);
int kbest_destroy_multi(kbest_multi *m);
int kbest_multi_size(const kbest_multi *m);
const char *kbest_multi_last_error(const kbest_multi *m);
int kbest_batch_f64_multi(kbest_multi *m, const kbest_opts *opts, int B, int maxRow, int maxCol, const int32_t *nRow,
                          const int32_t *nCol, const double *cost, int k, int32_t *row4col, int32_t *col4row, double *gain,
                          int32_t *nf);
/*
 * The same with the sharding mode explicit.  KBEST_MULTI_BATCH: as above.  KBEST_MULTI_SUBTREE (few large matrices; the
 * north star's "per-rank top-k into a global k-best heap"): every device receives ALL B matrices; shard s of nShard
 * (0 = one per device; more than devices: dealt round robin, a device runs its shards one after the other) expands only
 * the root's children on columns c % nShard == s (reference partition: split, shortestPathCPP.cpp:455-532) and enumerates
 * its own k best.  The exchange is gains first (numRow <= 127): ONE all-gather of every shard's top-k COSTS (gain[k] + nf), the merge
 * into the global k-best heap on every device (kbest_merge_gains_f64_dev), and ONE sum all-reduce of the byte table that holds every
 * winner's row at its merged position -- 8 k S + k numCol bytes per matrix instead of (8 + 4 numCol) k S.  A call in which two
 * candidates have exactly the same gain (integer-like costs: their order is the assignments'), and problems of more than 127 rows,
 * all-gather the whole per-shard lists and merge those (kbest_merge_topk[_i8]_f64_dev).  Results are those of the batch mode for
 * tie-free costs (exact ties: ordered by the assignment); col4row, which is not part of the exchange, is returned as the inverse of
 * row4col with -1 for rows without a real column.  opts->root_col_offset / stride must be unset.
 */
#define KBEST_MULTI_BATCH 0
#define KBEST_MULTI_SUBTREE 1
int kbest_batch_f64_multi_ex(kbest_multi *m, const kbest_opts *opts, int mode, int nShard, int B, int maxRow, int maxCol,
                             const int32_t *nRow, const int32_t *nCol, const double *cost, int k, int32_t *row4col,
                             int32_t *col4row, double *gain, int32_t *nf);
/* 1 when every device holds the same global table after the last kbest_batch_f64_multi[_ex] call, 0 when not (test aid). */
int kbest_multi_tables_agree(kbest_multi *m);
/* Bytes that ARRIVE at one device in the exchanges of the last kbest_batch_f64_multi[_ex] call: an all-gather of b bytes per device
 * brings (G - 1) b, the ring all-reduce of the n-byte table of subtree mode 2 n (G - 1) / G; *path (optional): 0 batch mode,
 * 1 subtree mode gains first, 2 subtree mode whole lists. */
long long kbest_multi_exchange_bytes(const kbest_multi *m, int *path);
/* KBEST_TIE_* flags of the problems of the last kbest_batch_f64_multi call (batch mode): copies min(n, cap) flags, returns n. */
int kbest_multi_last_tie_flags(kbest_multi *m, int32_t *flags, int cap);
/*
 * Host timeline of the last kbest_batch_f64_multi[_ex] call: out[g * KBEST_MULTI_STAMPS + i], seconds since the call was
 * entered, for device g: [0] its worker thread started, [1] its first upload was issued, [2] its first kernel was issued,
 * [3] it was fed (batch mode: its own results are back in the caller's tables as well), [4] the exchange was issued, [5] done.
 * Every device is fed and read back by a thread of its own, so no device waits for another one's copies; this is the
 * evidence.  Returns the number of devices written (at most capDevices).
 */
#define KBEST_MULTI_STAMPS 6
int kbest_multi_timeline(const kbest_multi *m, double *out, int capDevices);

#ifdef __cplusplus
}
#endif
#endif /* KBEST_C_H */
