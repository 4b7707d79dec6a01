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
    return 0;
}
