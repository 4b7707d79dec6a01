// ref_caller_main.cpp -- main() of oracle/_ref/ref_caller_on_engine: the reference's OWN caller code on the GPU engine.
//
// TEST INFRASTRUCTURE ONLY (see oracle/kbest_oracle.c for the rules).  This file holds no reference code.
// oracle/Makefile links it with ref_assign_shim.cpp -- the verbatim slices assignment.cpp:439-542, 547-683, 835-964
// compiled against the reference's own shortestPathCPP.hpp -- and with probabilisticsemslam_amd/libkbest_amd.so IN PLACE
// OF shortestPathCPP.cpp.  The reference's call sites assignment.cpp:594 (kBest2DCutoff) and :880 (kBest2D) thereby
// run unchanged, with the reference's own ScratchSpace (hpp:73-142) and caller-allocated tables, and land in the
// engine's drop-in entry points (include/kbest_shims.hpp): the link-level replacement INTEGRATION.md describes, proven
// with the reference's object code instead of a look-alike caller.  conditionCosts / assignmentProb / bruteForceProb
// executed here are the executable's own (the reference's), not the engine's fused versions of the same names.
//
//   ref_caller_on_engine in.bin out.bin
//   in.bin : int32 nCases; per case int32 {brute, nL, nM, k} + (nL+nM)*nM doubles (raw block, column-major)
//   out.bin: per case int32 {goodRows, width} + nM*width doubles (assignmentProb of the conditioned block)
//            [+ int32 width2 + nM*width2 doubles (bruteForceProb) when brute]
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" {
int ref_assignment_prob(const double *cost, int nL, int nM, int k, double *probs);
int ref_brute_force_prob(const double *cost, int nL, int nM, double *probs);
int ref_condition_costs(const double *cost, int nRows, int nCols, double *out, int64_t *rowIdx);
}

static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!in || !out) { perror("open"); return 2; }
    int32_t n = 0;
    if (!rd(in, &n, 4)) return 2;
    for (int32_t c = 0; c < n; c++) {
        int32_t h[4];
        if (!rd(in, h, 16)) return 2;
        const int brute = h[0], nL = h[1], nM = h[2], k = h[3], nR = nL + nM;
        std::vector<double> raw((size_t)nR * nM), cond((size_t)nR * nM);
        std::vector<int64_t> idx(nR);
        if (!rd(in, raw.data(), raw.size() * 8)) return 2;
        const int good = ref_condition_costs(raw.data(), nR, nM, cond.data(), idx.data());
        const int condL = good - nM;
        std::vector<double> p((size_t)nM * (size_t)(good * nM + 1));
        int32_t w[2] = {good, ref_assignment_prob(cond.data(), condL, nM, k, p.data())};
        fwrite(w, 4, 2, out);
        fwrite(p.data(), 8, (size_t)nM * w[1], out);
        if (brute) {
            int32_t w2 = ref_brute_force_prob(cond.data(), condL, nM, p.data());
            fwrite(&w2, 4, 1, out);
            fwrite(p.data(), 8, (size_t)nM * w2, out);
        }
    }
    fclose(out);
    fclose(in);
    return 0;
}
