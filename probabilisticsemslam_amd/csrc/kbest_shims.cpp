// kbest_shims.cpp -- reference-named C++ entry points (include/kbest_shims.hpp)
// forwarding to the C ABI with B = 1 on a lazily created process-global context.
#include "kbest_shims.hpp"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "kbest_c.h"

namespace {

// KBEST_SHIM_REFERENCE_ORDER, exact ties in the drop-in (kbest_c.h, "Order of exact ties").  Unset (-1): kBest2D / kBest2DCutoff answer
// as the reference does -- the synchronous entry's default: a problem with an exact tie among its k + 1 best gains runs again on the
// reference-order kernel --, assignmentProb / bruteForceProb weigh the engine's choice among a tied level.  2: those too weigh the
// reference's own k best (kbest_set_reference_order(ctx, 2): frames with a tie at slot k run again).  1: everything on the
// reference-order kernel.  0: the engine's own rule everywhere (KBEST_FLAG_CANONICAL_TIES).
int shim_reference_order()
{
    static const int on = [] { const char *e = getenv("KBEST_SHIM_REFERENCE_ORDER"); return (e && *e) ? (*e == '2' ? 2 : (*e == '0' ? 0 : 1)) : -1; }();
    return on;
}

kbest_ctx *global_ctx()
{
    static kbest_ctx *ctx = nullptr;
    static std::once_flag once;
    static int rc = KBEST_OK;
    std::call_once(once, [] {
        rc = kbest_create(&ctx, 0);
        if (rc == KBEST_OK && shim_reference_order() > 0) kbest_set_reference_order(ctx, shim_reference_order());  // (assignmentProb / bruteForceProb as well)
    });
    if (rc != KBEST_OK || !ctx) throw std::runtime_error(std::string("kbest engine: ") + kbest_strerror(rc));
    return ctx;
}

void check(kbest_ctx *ctx, int rc)
{
    if (rc != KBEST_OK)
        throw std::runtime_error(std::string("kbest engine: ") + kbest_strerror(rc) + " (" + kbest_last_error(ctx) + ")");
}

size_t kbest_one(size_t k, size_t numRow, size_t numCol, bool maximize, const double *C, bool useCut, double cutoff,
                 ptrdiff_t *col4rowBest, ptrdiff_t *row4colBest, double *gainBest)
{
    kbest_ctx *ctx = global_ctx();
    kbest_opts o;
    kbest_default_opts(&o);
    o.maximize = maximize;
    o.use_cutoff = useCut;
    o.cutoff = cutoff;
    // By default a problem with an exact tie among its k + 1 best gains runs again on the reference-order kernel: the reference's own
    // answer (kbest_c.h).  KBEST_SHIM_REFERENCE_ORDER=1: every problem on that kernel (col4row on padded columns as the reference names
    // them too; slower); =0: the engine's own rule on exact ties
    if (shim_reference_order() == 1) o.flags |= KBEST_FLAG_REFERENCE_ORDER;
    else if (shim_reference_order() == 0) o.flags |= KBEST_FLAG_CANONICAL_TIES;
    std::vector<int32_t> r4c(k * numCol), c4r(k * numRow);
    int32_t nf = 0;
    check(ctx, kbest_batch_f64(ctx, &o, 1, (int)numRow, (int)numCol, nullptr, nullptr, C, nullptr, (int)k, r4c.data(),
                               c4r.data(), gainBest, &nf, nullptr));
    if (nf < 0) throw std::runtime_error("kbest engine: internal error");
    // the reference also writes the slot it breaks on under a cutoff (cpp:705-718); callers only read
    // the first `nf` slots, which is what is widened here
    const size_t n = (size_t)nf;
    for (size_t i = 0; i < n * numRow; i++) col4rowBest[i] = c4r[i];
    for (size_t i = 0; i < n * numCol; i++) row4colBest[i] = r4c[i];
    return n;
}

}  // namespace

MurtyHyp::MurtyHyp(const size_t numRow, const size_t numCol)
{
    const size_t bytes = numRow * sizeof(ptrdiff_t) + numCol * sizeof(ptrdiff_t) + numCol * sizeof(double) +
                         numRow * sizeof(double) + numRow * sizeof(bool);
    buffer = new char[bytes];
    char *p = buffer;
    u = reinterpret_cast<double *>(p);            p += numCol * sizeof(double);
    v = reinterpret_cast<double *>(p);            p += numRow * sizeof(double);
    col4row = reinterpret_cast<ptrdiff_t *>(p);   p += numRow * sizeof(ptrdiff_t);
    row4col = reinterpret_cast<ptrdiff_t *>(p);   p += numCol * sizeof(ptrdiff_t);
    forbiddenActiveRows = reinterpret_cast<bool *>(p);
    gain = 0.0;
    activeCol = 0;
    solved = false;
}

void ScratchSpace::init(const size_t numRow, const size_t numCol)
{
    // Same capacity contract as the reference (C must hold numRow*numCol doubles,
    // shortestPathCPP.hpp:100-119); the engine itself works in device memory.
    delete[] buffer;
    const size_t bytes = numRow * numCol * sizeof(double) + numRow * sizeof(double) + numCol * sizeof(size_t) +
                         numRow * (sizeof(size_t) + 2 * sizeof(ptrdiff_t) + 2 * sizeof(bool));
    buffer = new char[bytes];
    char *p = buffer;
    C = reinterpret_cast<double *>(p);                 p += numRow * numCol * sizeof(double);
    shortestPathCost = reinterpret_cast<double *>(p);  p += numRow * sizeof(double);
    ScannedColIdx = reinterpret_cast<size_t *>(p);     p += numCol * sizeof(size_t);
    pred = reinterpret_cast<size_t *>(p);              p += numRow * sizeof(size_t);
    Row2ScanParent = reinterpret_cast<ptrdiff_t *>(p); p += numRow * sizeof(ptrdiff_t);
    Row2Scan = reinterpret_cast<ptrdiff_t *>(p);       p += numRow * sizeof(ptrdiff_t);
    ScannedRows = reinterpret_cast<bool *>(p);         p += numRow * sizeof(bool);
    forbiddenActiveRows = reinterpret_cast<bool *>(p);
    toCut = false;
}

size_t kBest2D(const size_t k, const size_t numRow, const size_t numCol, const bool maximize, const double *C,
               ScratchSpace &, ptrdiff_t *col4rowBest, ptrdiff_t *row4colBest, double *gainBest)
{
    return kbest_one(k, numRow, numCol, maximize, C, false, 0.0, col4rowBest, row4colBest, gainBest);
}

size_t kBest2DCutoff(const size_t k, const size_t numRow, const size_t numCol, const bool maximize, const double *C,
                     ScratchSpace &workMem, ptrdiff_t *col4rowBest, ptrdiff_t *row4colBest, double *gainBest,
                     double cutoff)
{
    workMem.toCut = true;  // cpp:650-651 (sticky, as in the reference)
    workMem.maximize = maximize;
    const size_t n = kbest_one(k, numRow, numCol, maximize, C, true, cutoff, col4rowBest, row4colBest, gainBest);
    if (n) workMem.cutoffGain = maximize ? gainBest[0] - cutoff : gainBest[0] + cutoff;
    return n;
}

namespace {

// Root solution of one rectangular problem with duals (kbest_assign_batch_f64), written into a MurtyHyp the way
// shortestPathCPP leaves it (cpp:134-139, 228-237).  Returns feasible (1/0).
int assign_one(const size_t numRow, const size_t numCol, const bool maximize, const bool shift, const size_t numCol4Gain,
               const double *C, MurtyHyp *sol)
{
    kbest_ctx *ctx = global_ctx();
    std::vector<int32_t> r4c(numCol), c4r(numRow);
    std::vector<double> u(numCol), v(numRow);
    double g = 0.0;
    int32_t ok = 0;
    check(ctx, kbest_assign_batch_f64(ctx, 1, (int)numRow, (int)numCol, nullptr, nullptr, C, nullptr, maximize ? 1 : 0,
                                      shift ? 1 : 0, (int)numCol4Gain, r4c.data(), c4r.data(), &g, u.data(), v.data(), &ok));
    sol->activeCol = 0;
    sol->solved = true;
    sol->gain = g;  // -1 when infeasible (cpp:200)
    if (!ok) return 0;
    for (size_t r = 0; r < numRow; r++) { sol->col4row[r] = c4r[r]; sol->v[r] = v[r]; sol->forbiddenActiveRows[r] = false; }
    for (size_t c = 0; c < numCol; c++) { sol->row4col[c] = r4c[c]; sol->u[c] = u[c]; }
    sol->forbiddenActiveRows[sol->row4col[0]] = true;  // cpp:235
    return 1;
}

}  // namespace

int assign2D(const size_t numRow, const size_t numCol, const bool maximize, const double *C, ScratchSpace &,
             MurtyHyp *problemSol)
{
    // cpp:735-762: makeCostMatrixSafe, numCol augmentations on the RECTANGULAR problem (unassigned rows stay -1),
    // gain un-shifted; problemSol also receives the dual variables, as in the reference
    return assign_one(numRow, numCol, maximize, true, numCol, C, problemSol);
}

int shortestPathCPP(MurtyHyp *problemSol, ScratchSpace &workMem, const size_t numRow, const size_t numCol,
                    const size_t numCol4Gain)
{
    // hpp:178-182 / cpp:119-238: the root LAP on the (already non-negative) matrix in workMem.C; returns 1 and
    // gain = -1 when infeasible (cpp:197-203)
    return assign_one(numRow, numCol, false, false, numCol4Gain, workMem.C, problemSol) ? 0 : 1;
}

void toProbs(std::vector<double> &costMatrix)
{
    kbest_ctx *ctx = global_ctx();
    check(ctx, kbest_to_probs_f64(ctx, costMatrix.data(), (int64_t)costMatrix.size()));
}

std::vector<std::vector<double>> assignmentProb(const std::vector<double> &costMatrix, size_t nL, size_t nM, size_t k)
{
    kbest_ctx *ctx = global_ctx();
    const int32_t l = (int32_t)nL, m = (int32_t)nM;
    const int64_t zero = 0;
    // the single-column path returns 1 x costMatrix.size() (assignment.cpp:557); otherwise nM x (nL+1)
    const size_t width = (nM == 1) ? costMatrix.size() : nL + 1;
    std::vector<double> flat(nM * (nL + 1), 0.0);
    check(ctx, kbest_weights_batch_f64(ctx, 1, &l, &m, costMatrix.data(), &zero, (int)k, flat.data(), &zero, nullptr));
    std::vector<std::vector<double>> probs(nM, std::vector<double>(width, 0.0));
    for (size_t c = 0; c < nM; c++)
        for (size_t j = 0; j <= nL; j++) probs[c][j] = flat[c * (nL + 1) + j];
    return probs;
}

std::vector<double> conditionCosts(const std::vector<double> &costs, size_t nRows, size_t nCols,
                                   std::vector<ptrdiff_t> &rowIdxOut)
{
    kbest_ctx *ctx = global_ctx();
    const int32_t nr = (int32_t)nRows, nc = (int32_t)nCols;
    const int64_t zero = 0;
    std::vector<double> out(nRows * nCols);
    std::vector<int32_t> ridx(nRows);
    int32_t good = 0;
    check(ctx, kbest_condition_costs_f64(ctx, 1, &nr, &nc, costs.data(), &zero, out.data(), &good, ridx.data(), (int)nRows));
    out.resize((size_t)good * nCols);
    std::vector<ptrdiff_t> idx(ridx.begin(), ridx.begin() + good);
    rowIdxOut.swap(idx);  // assignment.cpp:523
    return out;
}

std::vector<std::vector<double>> bruteForceProb(const std::vector<double> &costMatrix, size_t nL, size_t nM)
{
    kbest_ctx *ctx = global_ctx();
    const size_t nRows = nL + nM;
    const int32_t l = (int32_t)nL, m = (int32_t)nM;
    const int64_t zero = 0;
    const size_t width = (nM == 1) ? costMatrix.size() : nL + 1;  // single-column path, assignment.cpp:843
    // number of assignments to ask for: the Minc-type bound of assignment.cpp:28-36, 858-868, capped at 20000
    size_t upperK = 1;
    if (nM > 1) {
        const double tau = 6.2831853071, n = (double)nRows, mm = (double)nM;
        double bound = std::pow(tau, (mm - n) / (2 * n)) * std::pow(n / mm, mm) * std::exp(mm / (12 * n * n) - 1 / (12 * mm + 1));
        for (size_t r = 0; r < nRows; r++) {
            size_t card = 1;
            for (size_t c = 0; c < nM; c++)
                if (costMatrix[c * nRows + r] < std::numeric_limits<double>::infinity()) card++;
            const double cn = (double)card;
            bound *= std::pow(tau * cn, 1.0 / (2.0 * cn)) * cn * std::exp(-1 + 1.0 / (12 * cn * cn));
        }
        upperK = (bound < 1.8e19) ? (size_t)bound + 1 : 20000;
        if (upperK > 20000) upperK = 20000;
    }
    std::vector<double> flat(nM * (nL + 1), 0.0);
    check(ctx, kbest_bruteforce_probs_batch_f64(ctx, 1, &l, &m, costMatrix.data(), &zero, (int)upperK, flat.data(), &zero,
                                                nullptr));
    std::vector<std::vector<double>> probs(nM, std::vector<double>(width, 0.0));
    for (size_t c = 0; c < nM; c++)
        for (size_t j = 0; j <= nL; j++) probs[c][j] = flat[c * (nL + 1) + j];
    return probs;
}

kbest_ctx *kbest_shims_context() { return global_ctx(); }
