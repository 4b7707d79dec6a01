// ref_assign_shim.cpp -- C entry points around VERBATIM SLICES of the reference's assignment.cpp.
//
// TEST INFRASTRUCTURE ONLY (see oracle/kbest_oracle.c for the rules).
// This file holds no reference code.  assignment.cpp as a whole cannot be compiled here (assignment.h:4-9
// pulls in Eigen, GTSAM, gtsam_quadrics and OpenCV, none of which are in the image), but the functions on the
// hot path are std-only.  oracle/Makefile therefore cuts these line ranges out of the reference where it lies --
//     constsUtils.h:10,18-21            inf_d, tic/toc
//     assignment.cpp:9-11               cutoff / apprxIter / tau
//     assignment.cpp:28-36              mincConstant, mincFactor
//     assignment.cpp:439-542            conditionCosts, toProbs
//     assignment.cpp:547-683            assignmentProb
//     assignment.cpp:835-964            bruteForceProb (to the end of the file; its last line has no newline)
// -- into a temporary file under /tmp (never into the repository) and hands its path to this translation unit as
// REF_ASSIGN_SLICE; the prelude below supplies the std headers those lines need.  The result,
// oracle/_ref/libref_assign.so (git-ignored, a binary), is linked against the unmodified reference solver and is
// used to record tests/golden/weights_golden.npz and to pin the oracle's weights functions.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <limits>
#include <numeric>
#include <vector>

#include "shortestPathCPP.hpp"

#include REF_ASSIGN_SLICE

extern "C" {

// assignmentProb (assignment.h:11).  probs: [nM][width] row-major, width = nL+1 (or costMatrix.size() when nM == 1).
int ref_assignment_prob(const double *cost, int nL, int nM, int k, double *probs)
{
    std::vector<double> c(cost, cost + (size_t)(nL + nM) * nM);
    const auto p = assignmentProb(c, (size_t)nL, (size_t)nM, (size_t)k);
    size_t o = 0;
    for (const auto &row : p)
        for (double x : row) probs[o++] = x;
    return (int)(p.empty() ? 0 : p[0].size());
}

// bruteForceProb (assignment.h:43)
int ref_brute_force_prob(const double *cost, int nL, int nM, double *probs)
{
    std::vector<double> c(cost, cost + (size_t)(nL + nM) * nM);
    const auto p = bruteForceProb(c, (size_t)nL, (size_t)nM);
    size_t o = 0;
    for (const auto &row : p)
        for (double x : row) probs[o++] = x;
    return (int)(p.empty() ? 0 : p[0].size());
}

// conditionCosts (assignment.h:26).  Returns goodRows; out holds goodRows x nCols column-major, rowIdx the row map.
int ref_condition_costs(const double *cost, int nRows, int nCols, double *out, int64_t *rowIdx)
{
    std::vector<double> c(cost, cost + (size_t)nRows * nCols);
    std::vector<ptrdiff_t> idx;
    const std::vector<double> r = conditionCosts(c, (size_t)nRows, (size_t)nCols, idx);
    std::copy(r.begin(), r.end(), out);
    for (size_t i = 0; i < idx.size(); i++) rowIdx[i] = (int64_t)idx[i];
    return (int)idx.size();
}

// toProbs (assignment.h:19), in place
void ref_to_probs(double *cost, int n)
{
    std::vector<double> c(cost, cost + n);
    toProbs(c);
    std::copy(c.begin(), c.end(), cost);
}

double ref_minc_constant(int n, int m) { return mincConstant((size_t)n, (size_t)m); }

}  // extern "C"
