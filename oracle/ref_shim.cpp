// ref_shim.cpp -- C entry points around the UNMODIFIED reference solver.
//
// TEST INFRASTRUCTURE ONLY (see oracle/kbest_oracle.c for the rules).
// This file holds no reference code: it includes the reference header from
// where it lies (-I$(REF), never copied) and forwards to the reference's own
// kBest2D / kBest2DCutoff / assign2D (shortestPathCPP.hpp:144-149, 204-212,
// 256-265).  oracle/Makefile compiles it together with
// $(REF)/shortestPathCPP.cpp into oracle/_ref/libref_kbest.so, which is
// git-ignored and is used (a) to generate tests/golden/*.npz, (b) to pin
// oracle/kbest_oracle.c, and (c) as bench.py's cpu_baseline kind "reference".
#include "shortestPathCPP.hpp"

#include <cstdint>
#include <vector>

extern "C" {

// One problem.  Outputs use the reference's own ptrdiff_t width.
int64_t ref_kbest2d(int64_t k, int64_t numRow, int64_t numCol, int maximize,
                    const double *C, int64_t *col4row, int64_t *row4col, double *gain)
{
    ScratchSpace ws;
    ws.init((size_t)numRow, (size_t)numRow);  // as every reference caller does (assignment.cpp:585)
    return (int64_t)kBest2D((size_t)k, (size_t)numRow, (size_t)numCol, maximize != 0, C, ws,
                            reinterpret_cast<ptrdiff_t *>(col4row),
                            reinterpret_cast<ptrdiff_t *>(row4col), gain);
}

int64_t ref_kbest2d_cutoff(int64_t k, int64_t numRow, int64_t numCol, int maximize,
                           const double *C, int64_t *col4row, int64_t *row4col, double *gain,
                           double cutoff)
{
    ScratchSpace ws;
    ws.init((size_t)numRow, (size_t)numRow);
    return (int64_t)kBest2DCutoff((size_t)k, (size_t)numRow, (size_t)numCol, maximize != 0, C, ws,
                                  reinterpret_cast<ptrdiff_t *>(col4row),
                                  reinterpret_cast<ptrdiff_t *>(row4col), gain, cutoff);
}

int ref_assign2d(int64_t numRow, int64_t numCol, int maximize, const double *C,
                 int64_t *col4row, int64_t *row4col, double *gain)
{
    ScratchSpace ws;
    ws.init((size_t)numRow, (size_t)numRow);
    MurtyHyp sol((size_t)numRow, (size_t)numRow);
    int ok = assign2D((size_t)numRow, (size_t)numCol, maximize != 0, C, ws, &sol);
    if (ok) {
        for (int64_t r = 0; r < numRow; r++) col4row[r] = sol.col4row[r];
        for (int64_t c = 0; c < numCol; c++) row4col[c] = sol.row4col[c];
        *gain = sol.gain;
    }
    return ok;
}

// assign2D (shift = 1) or shortestPathCPP on ws.C as it is (shift = 0), with everything the reference leaves in the
// MurtyHyp: col4row, row4col, gain, u (per column), v (per row).  Returns 1 = solved, 0 = infeasible.
int ref_assign2d_ex(int64_t numRow, int64_t numCol, int maximize, int shift, int64_t gainCols, const double *C,
                    int64_t *col4row, int64_t *row4col, double *gain, double *u, double *v)
{
    ScratchSpace ws;
    ws.init((size_t)numRow, (size_t)numRow);
    MurtyHyp sol((size_t)numRow, (size_t)numRow);
    int ok;
    if (shift) {
        ok = assign2D((size_t)numRow, (size_t)numCol, maximize != 0, C, ws, &sol);
    } else {
        for (int64_t i = 0; i < numRow * numCol; i++) ws.C[i] = C[i];
        ok = shortestPathCPP(&sol, ws, (size_t)numRow, (size_t)numCol, (size_t)(gainCols > 0 ? gainCols : numCol)) == 0;
    }
    *gain = sol.gain;
    if (ok) {
        for (int64_t r = 0; r < numRow; r++) { col4row[r] = sol.col4row[r]; v[r] = sol.v[r]; }
        for (int64_t c = 0; c < numCol; c++) { row4col[c] = sol.row4col[c]; u[c] = sol.u[c]; }
    }
    return ok;
}

// B equally-shaped problems packed back to back, one kBest2D call each with a
// fresh ScratchSpace and fresh outputs -- exactly how assignmentProb drives the
// solver (assignment.cpp:583-594).  Single thread.  Used for the CPU baseline.
int64_t ref_kbest2d_batch(int64_t B, int64_t k, int64_t numRow, int64_t numCol, int maximize,
                          const double *C, int64_t *col4row, int64_t *row4col, double *gain,
                          int64_t *nf)
{
    int64_t total = 0;
    for (int64_t b = 0; b < B; b++) {
        int64_t n = ref_kbest2d(k, numRow, numCol, maximize, C + b * numRow * numCol,
                                col4row + b * k * numRow, row4col + b * k * numCol, gain + b * k);
        nf[b] = n;
        total += n;
    }
    return total;
}

}  // extern "C"
