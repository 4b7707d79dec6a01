// A caller written the way the reference's own callers are (assignment.cpp:583-594, 742-750): ScratchSpace,
// caller-owned output arrays, kBest2D / kBest2DCutoff / assign2D / assignmentProb / conditionCosts / bruteForceProb by name.
// Compiled against include/kbest_shims.hpp and linked to libkbest_amd.so; prints everything in hex floats so
// that tests/test_gpu_parity.py can compare bit-for-bit with the checker.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "kbest_shims.hpp"

static uint64_t sm_state;
static double u01()
{
    sm_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = sm_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

int main(int argc, char **argv)
{
    const size_t N = argc > 1 ? atoi(argv[1]) : 8, M = argc > 2 ? atoi(argv[2]) : 8, k = argc > 3 ? atoi(argv[3]) : 10;
    sm_state = argc > 4 ? strtoull(argv[4], nullptr, 0) : 12345;
    std::vector<double> C(N * M);
    for (auto &x : C) x = u01();

    ScratchSpace workMem;
    workMem.init(N, N);
    std::vector<ptrdiff_t> rowAssignments(N * k), colAssignments(M * k);
    std::vector<double> costs(k);
    size_t nf = kBest2D(k, N, M, false, C.data(), workMem, rowAssignments.data(), colAssignments.data(), costs.data());
    printf("kBest2D nf %zu\n", nf);
    for (size_t s = 0; s < nf; s++) {
        printf("g %a r4c", costs[s]);
        for (size_t c = 0; c < M; c++) printf(" %td", colAssignments[s * M + c]);
        printf(" c4r");
        for (size_t r = 0; r < N; r++) printf(" %td", rowAssignments[s * N + r]);
        printf("\n");
    }
    nf = kBest2DCutoff(k, N, M, false, C.data(), workMem, rowAssignments.data(), colAssignments.data(), costs.data(), 0.1);
    printf("kBest2DCutoff nf %zu toCut %d\n", nf, (int)workMem.toCut);
    for (size_t s = 0; s < nf; s++) printf("g %a\n", costs[s]);

    MurtyHyp sol(N, N);
    int ok = assign2D(N, M, false, C.data(), workMem, &sol);
    printf("assign2D ok %d g %a r4c", ok, sol.gain);
    for (size_t c = 0; c < M; c++) printf(" %td", sol.row4col[c]);
    printf("\n");

    // weights on a small gated problem: 6 landmarks, 3 measurements (SURVEY 8(d) C5 generator, one frame)
    const size_t nL = 6, nM = 3, nR = nL + nM;
    std::vector<double> W(nR * nM, std::numeric_limits<double>::infinity());
    sm_state = 0xC0FFEE;
    for (size_t c = 0; c < nM; c++) {
        for (size_t r = 0; r < nL; r++) {
            double t = u01();
            if (t < 3.0 / nL || r == c) { double a = u01(), b = u01(); W[c * nR + r] = 12.0 * a * b; }
            else { double a = u01(); W[c * nR + r] = 60.0 + 400.0 * a; }
        }
        W[c * nR + nL + c] = 10.0;
    }
    std::vector<ptrdiff_t> rowIdx;
    std::vector<double> cond = conditionCosts(W, nR, nM, rowIdx);
    printf("conditionCosts rows %zu idx", rowIdx.size());
    for (auto i : rowIdx) printf(" %td", i);
    printf("\n");
    const size_t condL = cond.size() / nM - nM;
    std::vector<std::vector<double>> p = assignmentProb(cond, condL, nM, 200);
    for (size_t m = 0; m < nM; m++) {
        printf("p");
        for (double x : p[m]) printf(" %a", x);
        printf("\n");
    }
    // the reference's "truth" generator on the same conditioned block (assignment.cpp:835-963)
    std::vector<std::vector<double>> q = bruteForceProb(cond, condL, nM);
    for (size_t m = 0; m < nM; m++) {
        printf("q");
        for (double x : q[m]) printf(" %a", x);
        printf("\n");
    }
    // toProbs (assignment.h:19) on the conditioned block
    std::vector<double> tp = cond;
    toProbs(tp);
    printf("t");
    for (double x : tp) printf(" %a", x);
    printf("\n");

    // assign2D on a rectangular maximise problem, with everything the reference leaves in the MurtyHyp
    // (cpp:735-762): 12 x 5, costs 10 * u01 - 3 of stream 201 (= tests/golden/assign_golden.npz "rect_12x5_max")
    {
        const size_t R = 12, Cc = 5;
        sm_state = 201;
        std::vector<double> A(R * Cc);
        for (auto &x : A) x = u01() * 10 - 3;
        ScratchSpace ws;
        ws.init(R, R);
        MurtyHyp h(R, R);
        int ok2 = assign2D(R, Cc, true, A.data(), ws, &h);
        printf("assign2D_rect ok %d g %a solved %d activeCol %zu r4c", ok2, h.gain, (int)h.solved, h.activeCol);
        for (size_t c = 0; c < Cc; c++) printf(" %td", h.row4col[c]);
        printf(" c4r");
        for (size_t r = 0; r < R; r++) printf(" %td", h.col4row[r]);
        printf(" u");
        for (size_t c = 0; c < Cc; c++) printf(" %a", h.u[c]);
        printf(" v");
        for (size_t r = 0; r < R; r++) printf(" %a", h.v[r]);
        printf(" forb");
        for (size_t r = 0; r < R; r++) printf(" %d", (int)h.forbiddenActiveRows[r]);
        printf("\n");
        // infeasible: one column all +inf ("infeasible_6x4": stream 208, entries 6..11)
        sm_state = 208;
        std::vector<double> I(6 * 4);
        for (auto &x : I) x = u01();
        for (int i = 6; i < 12; i++) I[i] = std::numeric_limits<double>::infinity();
        MurtyHyp hi(6, 6);
        ws.init(6, 6);
        printf("assign2D_infeasible ok %d\n", assign2D(6, 4, false, I.data(), ws, &hi));
        // shortestPathCPP on workMem.C as it is, gain over 4 of 10 columns ("spc_10x10_g4": stream 209)
        sm_state = 209;
        ws.init(10, 10);
        for (size_t i = 0; i < 100; i++) ws.C[i] = u01();
        MurtyHyp hs(10, 10);
        int rc = shortestPathCPP(&hs, ws, 10, 10, 4);
        printf("shortestPathCPP rc %d g %a r4c", rc, hs.gain);
        for (size_t c = 0; c < 10; c++) printf(" %td", hs.row4col[c]);
        printf(" u");
        for (size_t c = 0; c < 10; c++) printf(" %a", hs.u[c]);
        printf("\n");
        // infeasible: column 1 all +inf ("spc_infeasible_4x4": stream 212)
        ws.init(4, 4);
        sm_state = 212;
        for (size_t i = 0; i < 16; i++) ws.C[i] = u01();
        for (int i = 4; i < 8; i++) ws.C[i] = std::numeric_limits<double>::infinity();
        MurtyHyp h4(4, 4);
        rc = shortestPathCPP(&h4, ws, 4, 4, 4);
        printf("shortestPathCPP_infeasible rc %d g %a\n", rc, h4.gain);
    }
    return 0;
}
