/*
 * kbest_oracle.c -- CPU restatement of the reference's k-best assignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product path (probabilisticsemslam_amd/csrc/*.hip behind include/kbest_c.h)
 * never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against tests/golden/*.npz, which tests/golden/gen_golden.py produced
 * by running the *unmodified* reference solver (/root/reference/
 * shortestPathCPP.cpp compiled by oracle/Makefile into oracle/_ref/) on the
 * same seeded inputs, and against the known-answer vectors of SURVEY.md 8(c).
 * The weights functions (orc_assignment_prob etc.) restate assignment.cpp.  The
 * file as a whole cannot be compiled here (it needs Eigen/GTSAM/OpenCV through
 * assignment.h:4-9), but its std-only line ranges can: oracle/Makefile cuts them
 * into oracle/_ref/libref_assign.so, and tests/golden/weights_golden.npz records
 * what THAT returns -- these restatements are pinned bit for bit by it
 * (tests/test_weights_golden.py), besides the SURVEY 8(c) weight KATs and the
 * exhaustive permutation / permanent identities of tests/test_weights.py.
 *
 * All file:line citations are relative to /root/reference.
 * Arithmetic: IEEE binary64, no reassociation, no contraction
 * (build with -O2 -ffp-contract=off, never -ffast-math).
 *
 * Data layout follows the reference: column-major cost matrix
 * C[row + col*numRow], rows >= cols, "u" is indexed by column and "v" by row
 * (shortestPathCPP.hpp:53-56).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INF (HUGE_VAL)

/* ---- per-call work counters (SURVEY 8(d) "work counts per problem") ---- */
typedef struct {
    int64_t children_solved; /* calls of the child solve (a7)                  */
    int64_t children_pushed; /* P: feasible, not cut children pushed (a8)      */
    int64_t dijkstra_steps;  /* iterations of the do{}while in child solves    */
    int64_t row_visits;      /* reduced-cost evaluations in child solves       */
    int64_t max_queue;       /* largest heap size seen                         */
    int64_t root_steps;      /* do{}while iterations in the root solve         */
} orc_stats;

/* ---- one Murty hypothesis (MurtyHyp, shortestPathCPP.hpp:22-65) ---- */
typedef struct {
    int32_t *col4row; /* [D] row -> col, -1 = unassigned */
    int32_t *row4col; /* [D] col -> row                  */
    double  *u;       /* [D] dual per column             */
    double  *v;       /* [D] dual per row                */
    uint8_t *forb;    /* [D] rows forbidden for activeCol */
    double   gain;
    int32_t  activeCol;
} orc_hyp;

static orc_hyp *hyp_new(int D)
{
    size_t bytes = sizeof(orc_hyp) + (size_t)D * (4 + 4 + 8 + 8 + 1) + 64;
    char *slab = (char *)malloc(bytes);
    orc_hyp *h = (orc_hyp *)slab;
    char *p = slab + ((sizeof(orc_hyp) + 15) & ~(size_t)15);
    h->u = (double *)p;        p += 8 * (size_t)D;
    h->v = (double *)p;        p += 8 * (size_t)D;
    h->col4row = (int32_t *)p; p += 4 * (size_t)D;
    h->row4col = (int32_t *)p; p += 4 * (size_t)D;
    h->forb = (uint8_t *)p;
    h->gain = 0.0;
    h->activeCol = 0;
    return h;
}

/* ---- scratch (ScratchSpace, shortestPathCPP.hpp:73-142) ---- */
typedef struct {
    int      D;
    double  *C;           /* [D*D] shifted, zero-padded cost copy            */
    double  *spc;         /* shortestPathCost[D]                             */
    int32_t *pred;        /* [D]                                             */
    int32_t *scannedCols; /* ScannedColIdx[D]                                */
    uint8_t *scannedRow;  /* ScannedRows[D]                                  */
    uint8_t *inScan;      /* membership form of Row2Scan[D]                  */
    uint8_t *forbStart;   /* workMem.forbiddenActiveRows[D]                  */
    int      toCut;       /* hpp:117, set by kBest2DCutoff cpp:650           */
    int      maximize;
    double   cutoffGain;
} orc_ws;

static void ws_init(orc_ws *w, int D)
{
    w->D = D;
    w->C = (double *)malloc(sizeof(double) * (size_t)D * (size_t)D);
    w->spc = (double *)malloc(sizeof(double) * (size_t)D);
    w->pred = (int32_t *)malloc(sizeof(int32_t) * (size_t)D);
    w->scannedCols = (int32_t *)malloc(sizeof(int32_t) * (size_t)D);
    w->scannedRow = (uint8_t *)malloc((size_t)D);
    w->inScan = (uint8_t *)malloc((size_t)D);
    w->forbStart = (uint8_t *)malloc((size_t)D);
    w->toCut = 0;
    w->maximize = 0;
    w->cutoffGain = 0.0;
}

static void ws_free(orc_ws *w)
{
    free(w->C); free(w->spc); free(w->pred); free(w->scannedCols);
    free(w->scannedRow); free(w->inScan); free(w->forbStart);
}

/* ScratchSpace::cutHyp, shortestPathCPP.hpp:130-131 */
static int ws_cut(const orc_ws *w, double gain)
{
    if (!w->toCut) return 0;
    return w->maximize ? (gain < w->cutoffGain) : (gain > w->cutoffGain);
}

/* calcGain, shortestPathCPP.cpp:59-80: serial left-to-right sum from 0.0 */
static double gain_of(const orc_ws *w, const orc_hyp *h, int numCol4Gain)
{
    double g = 0.0;
    for (int c = 0; c < numCol4Gain; c++)
        g = g + w->C[(size_t)c * (size_t)w->D + (size_t)h->row4col[c]];
    return g;
}

/*
 * One shortest augmenting path from column `start`, then the dual update and
 * the path flip.  Restates the shared body of shortestPathCPP (cpp:146-230)
 * and shortestPathUpdateCPP (cpp:283-358) plus updateDualAndAugment
 * (cpp:82-117).
 *
 * The reference keeps the rows still to be scanned as an ascending index
 * list (Row2Scan) and memmoves the chosen row out (cpp:215, 345); because the
 * list is always ascending (cpp:155-157 for the root, qsort at cpp:486 for a
 * split) that is the same as walking rows 0..D-1 with a membership flag,
 * which is what w->inScan is.  Tie-breaks: strict '<' (cpp:185,191,314,320),
 * so the earliest scanned column keeps pred and the lowest row index wins
 * the arg-min.
 *
 * forbStart != NULL: rows flagged there are skipped, but only while the
 * start column itself is being scanned (cpp:310).
 * Returns 1 if infeasible (cpp:197-203, 327-334), else 0.
 */
static int augment(orc_ws *w, orc_hyp *h, int start, const uint8_t *forbStart,
                   int64_t *steps, int64_t *visits)
{
    const int D = w->D;
    int nScannedCols = 0, sink = -1, cur = start;
    double delta = 0.0;

    memset(w->scannedRow, 0, (size_t)D);
    for (int r = 0; r < D; r++) w->spc[r] = ORC_INF;

    do {
        double minVal = ORC_INF;
        int closest = -1;
        w->scannedCols[nScannedCols++] = cur;
        if (steps) (*steps)++;
        for (int r = 0; r < D; r++) {
            if (!w->inScan[r]) continue;
            if (forbStart && cur == start && forbStart[r]) continue;
            /* cpp:183 / cpp:313: ((delta + C) - u) - v, left to right */
            double rc = delta + w->C[(size_t)r + (size_t)cur * (size_t)D]
                        - h->u[cur] - h->v[r];
            if (visits) (*visits)++;
            if (rc < w->spc[r]) { w->pred[r] = cur; w->spc[r] = rc; }
            if (w->spc[r] < minVal) { minVal = w->spc[r]; closest = r; }
        }
        if (minVal == ORC_INF) return 1;
        w->scannedRow[closest] = 1;
        w->inScan[closest] = 0;
        delta = w->spc[closest];
        if (h->col4row[closest] == -1) sink = closest;
        else cur = h->col4row[closest];
    } while (sink == -1);

    /* updateDualAndAugment, cpp:82-117 */
    h->u[start] = h->u[start] + delta;
    for (int i = 1; i < nScannedCols; i++) {
        int c = w->scannedCols[i];
        h->u[c] = h->u[c] + delta - w->spc[h->row4col[c]];
    }
    for (int r = 0; r < D; r++)
        if (w->scannedRow[r]) h->v[r] = h->v[r] - delta + w->spc[r];
    {
        int r = sink, c;
        do {
            c = w->pred[r];
            h->col4row[r] = c;
            int nxt = h->row4col[c];
            h->row4col[c] = r;
            r = nxt;
        } while (c != start);
    }
    return 0;
}

/* shortestPathCPP, cpp:119-238 (root LAP; numRow == numCol == D here,
 * numCol4Gain = M).  Returns 1 if infeasible. */
static int root_solve(orc_ws *w, orc_hyp *h, int numCol4Gain, orc_stats *st)
{
    const int D = w->D;
    for (int i = 0; i < D; i++) {
        h->col4row[i] = -1; h->row4col[i] = -1;
        h->u[i] = 0.0; h->v[i] = 0.0; h->forb[i] = 0;
    }
    h->activeCol = 0;
    for (int c = 0; c < D; c++) {
        memset(w->inScan, 1, (size_t)D);
        if (augment(w, h, c, NULL, st ? &st->root_steps : NULL, NULL)) {
            h->gain = -1.0;
            return 1;
        }
    }
    h->gain = gain_of(w, h, numCol4Gain);
    h->forb[h->row4col[0]] = 1; /* cpp:235 */
    return 0;
}

/* shortestPathUpdateCPP, cpp:240-365.  w->inScan and w->forbStart are set by
 * the caller (split).  Child gets gain = -1 if infeasible. */
static orc_hyp *child_solve(orc_ws *w, const orc_hyp *parent, int cur,
                            int numVarCol, orc_stats *st)
{
    const int D = w->D;
    orc_hyp *h = hyp_new(D);
    h->activeCol = cur;
    memcpy(h->row4col, parent->row4col, 4 * (size_t)D);
    memcpy(h->col4row, parent->col4row, 4 * (size_t)D);
    memcpy(h->u, parent->u, 8 * (size_t)D);
    memcpy(h->v, parent->v, 8 * (size_t)D);
    memcpy(h->forb, w->forbStart, (size_t)D);  /* cpp:274 */
    h->col4row[h->row4col[cur]] = -1;          /* cpp:277 */
    h->row4col[cur] = -1;                      /* cpp:278 */
    if (st) st->children_solved++;
    if (augment(w, h, cur, w->forbStart, st ? &st->dijkstra_steps : NULL,
                st ? &st->row_visits : NULL)) {
        h->gain = -1.0;
        return h;
    }
    h->gain = gain_of(w, h, numVarCol);
    h->forb[h->row4col[cur]] = 1;              /* cpp:362 */
    return h;
}

/* ---- binary heap with the exact sift rules of libstdc++'s
 * std::priority_queue (bits/stl_heap.h: __push_heap / __adjust_heap, GCC 11),
 * which is what cpp:574 instantiates with pMurtyHyp::operator< (cpp:35-37:
 * a < b  <=>  a.gain > b.gain).  Restating those rules keeps the pop order
 * of equal-gain hypotheses identical to the compiled reference. ---- */
typedef struct { orc_hyp **a; int n, cap; } orc_heap;

static int heap_less(const orc_hyp *x, const orc_hyp *y) { return x->gain > y->gain; }

static void heap_sift_up(orc_heap *q, int hole, int top, orc_hyp *val)
{
    int parent = (hole - 1) / 2;
    while (hole > top && heap_less(q->a[parent], val)) {
        q->a[hole] = q->a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q->a[hole] = val;
}

static void heap_push(orc_heap *q, orc_hyp *h)
{
    if (q->n == q->cap) {
        q->cap = q->cap ? 2 * q->cap : 64;
        q->a = (orc_hyp **)realloc(q->a, sizeof(orc_hyp *) * (size_t)q->cap);
    }
    q->n++;
    heap_sift_up(q, q->n - 1, 0, h);
}

static orc_hyp *heap_pop(orc_heap *q)
{
    orc_hyp *top = q->a[0];
    int len = q->n - 1;            /* heap length after removing the top */
    orc_hyp *val = q->a[len];
    q->n = len;
    if (len == 0) return top;
    int hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (heap_less(q->a[child], q->a[child - 1])) child--;
        q->a[hole] = q->a[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q->a[hole] = q->a[child - 1];
        hole = child - 1;
    }
    heap_sift_up(q, hole, 0, val);
    return top;
}

/* split, cpp:455-532 */
static void split(orc_ws *w, const orc_hyp *parent, orc_heap *q, int numVarCol,
                  orc_stats *st)
{
    const int D = w->D;
    const int a = parent->activeCol;
    for (int c = a; c < numVarCol; c++) {
        /* rows still owned by columns >= c of the parent (cpp:480-488 for
         * c == a; cpp:506-508, 512, 525-527 remove one row per later child) */
        memset(w->inScan, 0, (size_t)D);
        for (int j = c; j < D; j++) w->inScan[parent->row4col[j]] = 1;
        if (c == a) {
            memcpy(w->forbStart, parent->forb, (size_t)D);   /* cpp:490 */
        } else {
            memset(w->forbStart, 0, (size_t)D);              /* cpp:510 */
            w->forbStart[parent->row4col[c]] = 1;            /* cpp:516 */
        }
        orc_hyp *h = child_solve(w, parent, c, numVarCol, st);
        if (h->gain == -1.0 || ws_cut(w, h->gain)) {         /* cpp:496, 521 */
            free(h);
        } else {
            heap_push(q, h);
            if (st) {
                st->children_pushed++;
                if (q->n > st->max_queue) st->max_queue = q->n;
            }
        }
    }
}

/* makeCostMatrixSafe (cpp:534-569) + zero padding (cpp:582-585, 663-666).
 * Returns CDelta (not yet multiplied by numCol). */
static double load_costs(orc_ws *w, const double *C, int N, int M, int maximize)
{
    const size_t n = (size_t)N * (size_t)M;
    double d = C[0];
    if (!maximize) {
        for (size_t i = 1; i < n; i++) if (C[i] < d) d = C[i];
        for (size_t i = 0; i < n; i++) w->C[i] = C[i] - d;
    } else {
        for (size_t i = 1; i < n; i++) if (d < C[i]) d = C[i];
        for (size_t i = 0; i < n; i++) w->C[i] = -C[i] + d;
    }
    for (size_t i = n; i < (size_t)N * (size_t)N; i++) w->C[i] = 0.0;
    return d;
}

static void emit(const orc_hyp *h, int N, int M, int slot, double CDelta,
                 int maximize, int32_t *col4rowBest, int32_t *row4colBest,
                 double *gainBest)
{
    memcpy(col4rowBest + (size_t)slot * (size_t)N, h->col4row, 4 * (size_t)N);
    memcpy(row4colBest + (size_t)slot * (size_t)M, h->row4col, 4 * (size_t)M);
    gainBest[slot] = maximize ? (-h->gain + CDelta) : (h->gain + CDelta);
}

/*
 * kBest2D (cpp:571-644) when use_cutoff == 0, kBest2DCutoff (cpp:646-733)
 * otherwise.  Outputs are int32 (the reference ABI is ptrdiff_t; values are
 * identical).  Returns the number of solutions found, 0 = infeasible.
 * col4rowBest entries >= M are the padded columns, verbatim.
 */
int orc_kbest(int k, int N, int M, int maximize, const double *C,
              int use_cutoff, double cutoff, int32_t *col4rowBest,
              int32_t *row4colBest, double *gainBest, orc_stats *st)
{
    orc_ws w;
    orc_heap q = {0, 0, 0};
    int sweep;
    if (st) memset(st, 0, sizeof(*st));
    ws_init(&w, N);
    if (use_cutoff) { w.toCut = 1; w.maximize = maximize; }  /* cpp:650-651 */

    double CDelta = load_costs(&w, C, N, M, maximize);
    CDelta = CDelta * (double)M;                             /* cpp:583 */

    orc_hyp *cur = hyp_new(N);
    if (root_solve(&w, cur, M, st)) {                        /* cpp:588-593 */
        free(cur);
        ws_free(&w);
        return 0;
    }
    emit(cur, N, M, 0, CDelta, maximize, col4rowBest, row4colBest, gainBest);
    if (use_cutoff)                                          /* cpp:680-686 */
        w.cutoffGain = maximize ? (cur->gain - cutoff) : (cur->gain + cutoff);
    heap_push(&q, cur);
    if (st) st->max_queue = 1;

    for (sweep = 1; sweep < k; sweep++) {                    /* cpp:607-634 */
        cur = heap_pop(&q);
        split(&w, cur, &q, M, st);
        free(cur);
        if (q.n == 0) break;
        emit(q.a[0], N, M, sweep, CDelta, maximize, col4rowBest, row4colBest,
             gainBest);
        if (use_cutoff) {                                    /* cpp:709-719 */
            if (!maximize) { if (gainBest[sweep] > gainBest[0] + cutoff) break; }
            else           { if (gainBest[sweep] < gainBest[0] - cutoff) break; }
        }
    }
    while (q.n) free(heap_pop(&q));
    free(q.a);
    ws_free(&w);
    return sweep;
}

/* Batched driver used by bench.py's cpu_baseline leg ("port" kind) and by the
 * tests: B problems of identical shape, packed back to back. */
int64_t orc_kbest_batch(int B, int k, int N, int M, int maximize,
                        const double *C, int use_cutoff, double cutoff,
                        int32_t *col4row, int32_t *row4col, double *gain,
                        int32_t *nf, int64_t *pushed)
{
    int64_t total = 0;
    for (int b = 0; b < B; b++) {
        orc_stats st;
        int n = orc_kbest(k, N, M, maximize, C + (size_t)b * N * M, use_cutoff,
                          cutoff, col4row + (size_t)b * k * N,
                          row4col + (size_t)b * k * M, gain + (size_t)b * k, &st);
        nf[b] = n;
        if (pushed) pushed[b] = st.children_pushed;
        total += n;
    }
    return total;
}

/* assign2D, cpp:735-762: N x M directly (no padding), single best. */
int orc_assign2d(int N, int M, int maximize, const double *C, int32_t *col4row,
                 int32_t *row4col, double *gain)
{
    /* The reference runs shortestPathCPP with numRow=N, numCol=M on an N x M
     * matrix (cpp:746-749).  The shared augment() above is written for the
     * square D x D case, so pad with zero columns: extra zero columns after
     * the real ones are never started from before column M-1 is assigned and
     * cannot change the first M augmentations, which is all assign2D does. */
    orc_ws w;
    ws_init(&w, N);
    double CDelta = load_costs(&w, C, N, M, maximize) * (double)M;
    orc_hyp *h = hyp_new(N);
    for (int i = 0; i < N; i++) {
        h->col4row[i] = -1; h->row4col[i] = -1; h->u[i] = 0; h->v[i] = 0; h->forb[i] = 0;
    }
    int bad = 0;
    for (int c = 0; c < M && !bad; c++) {
        memset(w.inScan, 1, (size_t)N);
        bad = augment(&w, h, c, NULL, NULL, NULL);
    }
    if (!bad) {
        double g = gain_of(&w, h, M);
        *gain = maximize ? (-g + CDelta) : (g + CDelta);
        memcpy(col4row, h->col4row, 4 * (size_t)N);
        memcpy(row4col, h->row4col, 4 * (size_t)M);
    }
    free(h);
    ws_free(&w);
    return bad ? 0 : 1;
}

/*
 * assign2D (cpp:735-762, shift = 1) or shortestPathCPP on an already non-negative matrix (cpp:119-238, shift = 0;
 * gain over the first gainCols columns, numCol4Gain of cpp:232), with the dual variables the reference leaves in the
 * MurtyHyp: u[M] per column, v[N] per row.  Returns 1 = solved, 0 = infeasible (then *gain = -1, cpp:200).
 */
int orc_assign2d_ex(int N, int M, int maximize, int shift, int gainCols, const double *C, int32_t *col4row,
                    int32_t *row4col, double *gain, double *u, double *v)
{
    orc_ws w;
    ws_init(&w, N);
    double CDelta = 0.0;
    if (shift) {
        CDelta = load_costs(&w, C, N, M, maximize) * (double)M;
    } else {
        for (size_t i = 0; i < (size_t)N * (size_t)M; i++) w.C[i] = C[i];
        for (size_t i = (size_t)N * (size_t)M; i < (size_t)N * (size_t)N; i++) w.C[i] = 0.0;
    }
    orc_hyp *h = hyp_new(N);
    for (int i = 0; i < N; i++) {
        h->col4row[i] = -1; h->row4col[i] = -1; h->u[i] = 0; h->v[i] = 0; h->forb[i] = 0;
    }
    int bad = 0;
    for (int c = 0; c < M && !bad; c++) {
        memset(w.inScan, 1, (size_t)N);
        bad = augment(&w, h, c, NULL, NULL, NULL);
    }
    if (!bad) {
        const double g = gain_of(&w, h, gainCols > 0 && gainCols < M ? gainCols : M);
        *gain = shift ? (maximize ? (-g + CDelta) : (g + CDelta)) : g;
        memcpy(col4row, h->col4row, 4 * (size_t)N);
        memcpy(row4col, h->row4col, 4 * (size_t)M);
        memcpy(u, h->u, 8 * (size_t)M);
        memcpy(v, h->v, 8 * (size_t)N);
    } else {
        *gain = -1.0;
    }
    free(h);
    ws_free(&w);
    return bad ? 0 : 1;
}

/* =====================  association weights (assignment.cpp)  ============ */

#define ORC_GATE 42.0 /* `cutoff`, assignment.cpp:9 (size_t 42, promoted) */

/* conditionCosts, assignment.cpp:439-525.  out must hold nRows*nCols doubles,
 * rowIdx nRows ints.  Returns goodRows; output is column-major
 * goodRows x nCols. */
int orc_condition_costs(const double *costs, int nRows, int nCols, double *out,
                        int32_t *rowIdx)
{
    double *colMin = (double *)malloc(sizeof(double) * (size_t)nCols);
    uint8_t *good = (uint8_t *)malloc((size_t)nRows);
    int goodRows = 0;
    for (int c = 0; c < nCols; c++) {                 /* :450-458 */
        colMin[c] = ORC_INF;
        for (int r = 0; r < nRows; r++)
            if (costs[(size_t)c * nRows + r] < colMin[c]) colMin[c] = costs[(size_t)c * nRows + r];
    }
    for (int r = 0; r < nRows; r++) {                 /* :462-474 */
        good[r] = 0;
        for (int c = 0; c < nCols; c++)
            if (costs[(size_t)c * nRows + r] <= colMin[c] + ORC_GATE) { good[r] = 1; goodRows++; break; }
    }
    int o = 0;
    for (int r = 0; r < nRows; r++) {                 /* :481-496 */
        if (!good[r]) continue;
        rowIdx[o] = r;
        for (int c = 0; c < nCols; c++) {
            double x = costs[(size_t)c * nRows + r];
            out[(size_t)c * goodRows + o] = (x <= colMin[c] + ORC_GATE) ? (x - colMin[c]) : ORC_INF;
        }
        o++;
    }
    free(colMin); free(good);
    return goodRows;
}

/* toProbs, assignment.cpp:527-542 */
void orc_to_probs(double *a, int n)
{
    double mn = a[0];
    for (int i = 1; i < n; i++) if (a[i] < mn) mn = a[i];
    for (int i = 0; i < n; i++) a[i] = (mn + ORC_GATE > a[i]) ? exp(mn - a[i]) : 0.0;
}

/* accumulate/normalise shared by assignmentProb (:616-648) and
 * bruteForceProb (:912-945).  probs is [nM][nL+1] row-major. */
static void weights_from_solutions(int nf, int nL, int nM, const int32_t *row4col,
                                   const double *gain, int gate, double *probs)
{
    for (int i = 0; i < nM * (nL + 1); i++) probs[i] = 0.0;
    double best = gain[0], total = 0.0;
    for (int s = 0; s < nf; s++) {
        if (gate && !(best + ORC_GATE > gain[s])) continue;  /* :622-626 */
        double p = exp(best - gain[s]);
        total += p;
        for (int c = 0; c < nM; c++) {
            int r = row4col[(size_t)s * nM + c];
            probs[(size_t)c * (nL + 1) + (r >= nL ? nL : r)] += p; /* :633-638 */
        }
    }
    double norm = 1.0 / total;                               /* :643 */
    for (int i = 0; i < nM * (nL + 1); i++) probs[i] *= norm;
}

/* the same accumulation for a GIVEN list of solutions (tests: the k best brought into the engine's one order of exact ties) */
void orc_weights_from_solutions(int nf, int nL, int nM, const int32_t *row4col, const double *gain, int gate, double *probs)
{
    weights_from_solutions(nf, nL, nM, row4col, gain, gate, probs);
}

/* nM == 1 fast path, assignment.cpp:554-570 / 840-856.  The reference returns
 * a 1 x (nL+nM) vector; entries above nL stay 0.  probs must hold nL+1. */
static void single_column(const double *cost, int nL, double *probs)
{
    double norm = 0.0;
    for (int i = 0; i <= nL; i++) {
        probs[i] = 0.0;
        if (cost[i] < ORC_GATE) { probs[i] = exp(-cost[i]); norm += probs[i]; }
    }
    norm = 1.0 / norm;
    for (int i = 0; i <= nL; i++) probs[i] = probs[i] * norm;
}

/* assignmentProb, assignment.cpp:547-683.  probs: [nM][nL+1] row-major.
 * Returns numFound (or -1 for the single-column fast path). */
int orc_assignment_prob(const double *cost, int nL, int nM, int k, double *probs)
{
    int nRows = nL + nM, nCols = nM;
    if (nM == 1) { single_column(cost, nL, probs); return -1; }
    int32_t *c4r = (int32_t *)malloc(4 * (size_t)nRows * k);
    int32_t *r4c = (int32_t *)malloc(4 * (size_t)nCols * k);
    double *g = (double *)malloc(8 * (size_t)k);
    int nf = orc_kbest(k, nRows, nCols, 0, cost, 1, ORC_GATE, c4r, r4c, g, NULL); /* :594 */
    weights_from_solutions(nf, nL, nM, r4c, g, 1, probs);
    free(c4r); free(r4c); free(g);
    return nf;
}

/* mincConstant / mincFactor, assignment.cpp:28-36 */
static double minc_constant(int ni, int mi)
{
    const double tau = 6.2831853071;
    double n = ni, m = mi;
    return pow(tau, (m - n) / (2 * n)) * pow(n / m, m) * exp(m / (12 * n * n) - 1 / (12 * m + 1));
}
static double minc_factor(int n)
{
    const double tau = 6.2831853071;
    return pow(tau * n, 1.0 / (2.0 * n)) * n * exp(-1 + 1.0 / (12 * n * n));
}

/* bruteForceProb, assignment.cpp:835-963.  Returns numFound; *upperK_out gets
 * the k it asked for (:868). */
int orc_brute_force_prob(const double *cost, int nL, int nM, double *probs, int *upperK_out)
{
    int nRows = nL + nM, nCols = nM;
    if (nM == 1) { single_column(cost, nL, probs); if (upperK_out) *upperK_out = 0; return -1; }
    double bound = minc_constant(nRows, nCols);              /* :858-867 */
    for (int r = 0; r < nRows; r++) {
        int card = 1;
        for (int c = 0; c < nCols; c++) if (cost[(size_t)c * nRows + r] < ORC_INF) card++;
        bound *= minc_factor(card);
    }
    size_t upperK = (size_t)bound + 1;                       /* :868 */
    if (!(bound < 1.8e19)) upperK = 20000;                   /* size_t cast of a huge double */
    if (upperK > 20000) upperK = 20000;
    int k = (int)upperK;
    if (upperK_out) *upperK_out = k;
    int32_t *c4r = (int32_t *)malloc(4 * (size_t)nRows * k);
    int32_t *r4c = (int32_t *)malloc(4 * (size_t)nCols * k);
    double *g = (double *)malloc(8 * (size_t)k);
    int nf = orc_kbest(k, nRows, nCols, 0, cost, 0, 0.0, c4r, r4c, g, NULL); /* :880 */
    weights_from_solutions(nf, nL, nM, r4c, g, 0, probs);
    free(c4r); free(r4c); free(g);
    return nf;
}

/* =====================  exact permanent (validator for config 5)  ========
 * nwPerm.cpp:217-231 (rectangular: pad with ones, divide by (m-n)!) and the
 * Ryser / Nijenhuis-Wilf inclusion-exclusion that nwPerm.cpp:251-332 runs in
 * Gray-code order.  "parity unpinned" for this one function: the reference
 * holds no test for it and Eigen is absent, so it is validated against an
 * O(n!) permutation sum instead (tests/test_weights.py).  Only used to check
 * weights on small exhaustive sub-problems. */
double orc_permanent_square(const double *A, int n) /* column-major n x n, n <= 30 */
{
    if (n == 0) return 1.0;
    double rowsum[32];
    for (int i = 0; i < n; i++) rowsum[i] = 0.0;
    double total = 0.0;
    uint32_t prev = 0;
    for (uint32_t i = 1; i < (1u << n); i++) {
        uint32_t gray = i ^ (i >> 1), diff = gray ^ prev;
        int col = 0;
        while (!((diff >> col) & 1u)) col++;
        double s = (gray & diff) ? 1.0 : -1.0;
        for (int r = 0; r < n; r++) rowsum[r] += s * A[(size_t)col * n + r];
        double prod = 1.0;
        for (int r = 0; r < n; r++) prod *= rowsum[r];
        int bits = __builtin_popcount(gray);
        total += (((n - bits) & 1) ? -1.0 : 1.0) * prod;
        prev = gray;
    }
    return total;
}

double orc_permanent(const double *A, int m, int n) /* column-major m x n */
{
    if (m == n) return orc_permanent_square(A, n);
    int d = m > n ? m : n;
    double *P = (double *)malloc(8 * (size_t)d * d);
    for (int i = 0; i < d * d; i++) P[i] = 1.0;
    for (int c = 0; c < n; c++)
        for (int r = 0; r < m; r++) P[(size_t)c * d + r] = A[(size_t)c * m + r];
    double scale = tgamma((double)abs(m - n) + 1.0);
    double p = orc_permanent_square(P, d) / scale;
    free(P);
    return p;
}

/* =====================  cost-matrix construction  ============== */

/* Solve S x = d for a symmetric positive (semi-)definite 3x3 S by LDL^T with diagonal pivoting -- the published
 * algorithm behind Eigen::LDLT ("robust Cholesky decomposition with pivoting", largest remaining diagonal entry
 * as pivot), which computeQuadricCostMatrix calls at assignment.cpp:717.  Eigen itself is a third-party dependency
 * that is absent from /root/reference (find_package(Eigen3), CMakeLists.txt:12, version unpinned), so this is
 * "parity unpinned" for the last bits: the tests hold it to 1e-12 relative against numpy's LU solve. */
static void ldlt3_solve(const double S[9], const double d[3], double x[3])
{
    double A[3][3], b[3];
    int perm[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++) { b[i] = d[i]; for (int j = 0; j < 3; j++) A[i][j] = S[i * 3 + j]; }
    double L[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, D[3];
    for (int k = 0; k < 3; k++) {
        int piv = k;
        for (int i = k + 1; i < 3; i++) if (fabs(A[i][i]) > fabs(A[piv][piv])) piv = i;
        if (piv != k) {  /* symmetric row/column swap */
            for (int j = 0; j < 3; j++) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
            for (int i = 0; i < 3; i++) { double t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
            for (int j = 0; j < k; j++) { double t = L[k][j]; L[k][j] = L[piv][j]; L[piv][j] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        D[k] = A[k][k];
        for (int i = k + 1; i < 3; i++) L[i][k] = A[i][k] / D[k];
        for (int i = k + 1; i < 3; i++)
            for (int j = k + 1; j < 3; j++) A[i][j] = A[i][j] - L[i][k] * D[k] * L[j][k];
    }
    double y[3], z[3];
    for (int i = 0; i < 3; i++) y[i] = b[perm[i]];
    for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) y[i] = y[i] - L[i][j] * y[j];
    for (int i = 0; i < 3; i++) y[i] = y[i] / D[i];
    for (int i = 2; i >= 0; i--) for (int j = i + 1; j < 3; j++) y[i] = y[i] - L[j][i] * y[j];
    for (int i = 0; i < 3; i++) z[perm[i]] = y[i];
    for (int i = 0; i < 3; i++) x[i] = z[i];
}

/* computeQuadricCostMatrix, assignment.cpp:705-722.  m1/cov1: nL landmarks (3 / 9 doubles each, cov row-major),
 * m2/cov2: nM measurements; out: (nL+nM) x nM column-major. */
void orc_quadric_costs(const double *m1, const double *cov1, int nL, const double *m2, const double *cov2, int nM,
                       double gate, double *out)
{
    const int nR = nL + nM;
    for (int i = 0; i < nR * nM; i++) out[i] = ORC_INF;
    for (int c = 0; c < nM; c++) {
        for (int r = 0; r < nL; r++) {
            double d[3], S[9], x[3];
            for (int i = 0; i < 3; i++) d[i] = m1[r * 3 + i] - m2[c * 3 + i];
            for (int i = 0; i < 9; i++) S[i] = cov1[r * 9 + i] + cov2[c * 9 + i];
            ldlt3_solve(S, d, x);
            out[(size_t)c * nR + r] = d[0] * x[0] + d[1] * x[1] + d[2] * x[2];
        }
        out[(size_t)c * nR + nL + c] = gate;
    }
}

/* boundBox::IoU, boundBox.h:62-75: `a` is *this (its xOffset is applied), `b` is `other`.  Boxes are
 * (xmin, ymin, xmax, ymax, xOffset). */
static double bb_iou(const double *a, const double *b)
{
    const double l = fmax(a[0] + a[4], b[0]), r = fmin(a[2] + a[4], b[2]);
    const double t = fmax(a[1], b[1]), bt = fmin(a[3], b[3]);
    if (l >= r || t >= bt) return 0.0;
    const double inter = (r - l) * (bt - t);
    const double areaA = (a[2] - a[0]) * (a[3] - a[1]), areaB = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / (areaA + areaB - inter);
}

/* computeBBCostMatrix, assignment.cpp:777-797: rows = right boxes + dummies, cols = left boxes. */
void orc_bb_costs(const double *bbL, int nL, const double *bbR, int nR, double gate, double *out)
{
    const int nRows = nR + nL;
    for (int i = 0; i < nRows * nL; i++) out[i] = -ORC_INF;
    for (int c = 0; c < nL; c++) {
        for (int r = 0; r < nR; r++) {
            const double i1 = bb_iou(bbR + 5 * r, bbL + 5 * c), i2 = bb_iou(bbL + 5 * c, bbR + 5 * r);
            out[(size_t)c * nRows + r] = i1 < i2 ? i1 : i2;  /* std::min(iou1, iou2) */
        }
        out[(size_t)c * nRows + nR + c] = gate;
    }
}

/* asgnBB, assignment.cpp:724-775: k = 1, maximize.  asg[nL]: index of the right box or -1. */
void orc_asgn_bb(const double *bbL, int nL, const double *bbR, int nR, double gate, int32_t *asg)
{
    for (int c = 0; c < nL; c++) asg[c] = -1;
    if (nL == 0 || nR == 0) return;
    const int nRows = nR + nL;
    double *cost = (double *)malloc(8 * (size_t)nRows * nL);
    int32_t *c4r = (int32_t *)malloc(4 * (size_t)nRows), *r4c = (int32_t *)malloc(4 * (size_t)nL);
    double g;
    orc_bb_costs(bbL, nL, bbR, nR, gate, cost);
    if (orc_kbest(1, nRows, nL, 1, cost, 0, 0.0, c4r, r4c, &g, NULL))
        for (int c = 0; c < nL; c++) if (r4c[c] < nR) asg[c] = r4c[c];
    free(cost); free(c4r); free(r4c);
}
