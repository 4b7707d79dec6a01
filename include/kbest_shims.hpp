// kbest_shims.hpp -- C++ drop-in declarations with the reference's own names
// and signatures, implemented on the MI355X engine (kbest_c.h) with B = 1.
//
// A caller written against the reference's shortestPathCPP.hpp / assignment.h
// keeps compiling against this header and links libkbest_amd.so instead of
// shortestPathCPP.o / the solver part of assignment.o:
//
//   kBest2D        shortestPathCPP.hpp:204-212   (callers: assignment.cpp:880)
//   kBest2DCutoff  shortestPathCPP.hpp:256-265   (callers: assignment.cpp:594)
//   assign2D       shortestPathCPP.hpp:144-149   (no caller in the reference)
//   shortestPathCPP shortestPathCPP.hpp:178-182  (callers: cpp:587, 668, 749)
//   toProbs        assignment.h:19               (callers: assignment.cpp:164)
//   assignmentProb assignment.h:11               (callers: assignment.cpp:66, comparison.cpp:194-222)
//   conditionCosts assignment.h:26               (callers: assignment.cpp:58, comparison.cpp:161)
//
// MurtyHyp / ScratchSpace keep the reference's public member names, types and
// declaration order (shortestPathCPP.hpp:22-65, 73-142) so that objects built
// by such a caller have the layout these functions expect.  The engine keeps
// its own device workspace, so ScratchSpace is only honoured as an interface:
// its flags are set the way kBest2DCutoff sets them (cpp:650-651), its buffers
// are not used.
//
// col4rowBest on zero-padded columns: for numRow > numCol the reference pads to a square, and WHICH padded column a
// left-over row lands on is an artefact of tie resolution among identical zero columns.  The
// engine's root starts from a column reduction and places left-over rows on the padded columns in ascending row order:
// values >= numCol in col4rowBest can differ from the reference's; values < numCol, row4colBest, gainBest and the return
// value are identical (the reference's own callers read only row4colBest, assignment.cpp:629).  KBEST_EXACT_ROOT=1 (or
// KBEST_FLAG_EXACT_ROOT) makes the root run the reference's own sequence of augmentations instead.
//
// Exact ties: hypotheses with exactly equal gains come back in the engine's one order -- (gain, row4col lexicographic), and
// the lexicographically first assignments of a gain level that straddles slot k (kbest_c.h, "Order of exact ties") -- not in
// the order of the reference's heap, which is an artefact (cpp:30-42, 574).  The flags of the last call are read with
// kbest_last_tie_flags on the shims' context (kbest_shims_context()).
//
// Error behaviour follows the reference: no exceptions from the solver, the
// return value is the number of solutions found and 0 means infeasible.  An
// engine failure (no GPU, unsupported size) cannot be expressed in that
// convention, so it throws std::runtime_error -- the product never silently
// falls back to a CPU path.
#ifndef KBEST_SHIMS_HPP
#define KBEST_SHIMS_HPP

#include <cstddef>
#include <vector>

class MurtyHyp {
private:
    char *buffer;

public:
    ptrdiff_t *col4row;
    ptrdiff_t *row4col;
    double gain;
    double *u;
    double *v;
    size_t activeCol;
    bool *forbiddenActiveRows;
    bool solved;

    MurtyHyp() : buffer(nullptr) {}
    MurtyHyp(const size_t numRow, const size_t numCol);
    ~MurtyHyp() { delete[] buffer; }
    MurtyHyp(const MurtyHyp &) = delete;
    MurtyHyp &operator=(const MurtyHyp &) = delete;
};

class ScratchSpace {
public:
    char *buffer;
    double *C;
    size_t *ScannedColIdx;
    bool *ScannedRows;
    size_t *pred;
    double *shortestPathCost;
    ptrdiff_t *Row2ScanParent;
    ptrdiff_t *Row2Scan;
    bool *forbiddenActiveRows;
    bool toCut;
    double cutoffGain;
    bool maximize;

    ScratchSpace() : buffer(nullptr), toCut(false) {}
    ScratchSpace(const size_t numRow, const size_t numCol) : buffer(nullptr) { init(numRow, numCol); }
    void init(const size_t numRow, const size_t numCol);
    ~ScratchSpace() { delete[] buffer; }
    inline bool cutHyp(double gain) const { return toCut ? (maximize ? gain < cutoffGain : gain > cutoffGain) : false; }
};

size_t kBest2D(const size_t k, const size_t numRow, const size_t numCol, const bool maximize, const double *C,
               ScratchSpace &workMem, ptrdiff_t *col4rowBest, ptrdiff_t *row4colBest, double *gainBest);

size_t kBest2DCutoff(const size_t k, const size_t numRow, const size_t numCol, const bool maximize, const double *C,
                     ScratchSpace &workMem, ptrdiff_t *col4rowBest, ptrdiff_t *row4colBest, double *gainBest,
                     double cutoff);

int assign2D(const size_t numRow, const size_t numCol, const bool maximize, const double *C, ScratchSpace &workMem,
             MurtyHyp *problemSol);

// shortestPathCPP.hpp:178-182 (cpp:119-238): the root LAP on workMem.C (numRow x numCol, already non-negative);
// fills problemSol (col4row, row4col, u, v, gain over the first numCol4Gain columns, forbiddenActiveRows);
// returns 1 with gain = -1 when infeasible, else 0
int shortestPathCPP(MurtyHyp *problemSol, ScratchSpace &workMem, const size_t numRow, const size_t numCol,
                    const size_t numCol4Gain);

std::vector<std::vector<double>> assignmentProb(const std::vector<double> &costMatrix, size_t nL, size_t nM, size_t k);

// assignment.h:43 (assignment.cpp:835-963): every assignment up to a Minc-type bound (at most 20000), no 42-gate
std::vector<std::vector<double>> bruteForceProb(const std::vector<double> &costMatrix, size_t nL, size_t nM);

// assignment.h:19 (assignment.cpp:527-542): exp(min - c) with the 42 gate, in place
void toProbs(std::vector<double> &costMatrix);

// assignment.h:26 (assignment.cpp:439-525)
std::vector<double> conditionCosts(const std::vector<double> &costs, size_t nRows, size_t nCols,
                                   std::vector<ptrdiff_t> &rowIdxOut);

// Not in the reference: the engine context behind the functions above (created on first use, GPU 0), for the entries of
// kbest_c.h that take one -- e.g. kbest_last_tie_flags after a call.
struct kbest_ctx;
kbest_ctx *kbest_shims_context();

#endif
