// proto_lazy.cpp -- DEV TOOL under tests/ (it links the CPU checker, which only tests may do): host model of the
// device algorithm used by csrc/kbest_engine.hip, to validate on the CPU that
// "lazy children + bounded sorted pool + early termination" returns exactly
// what the reference enumeration returns, and to count how much Dijkstra work
// the pruning removes.  Build: g++ -O2 -ffp-contract=off -std=c++17 tests/dev/proto_lazy.cpp oracle/libkbest_oracle.so -Wl,-rpath,$PWD/oracle
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" {
typedef struct { int64_t children_solved, children_pushed, dijkstra_steps, row_visits, max_queue, root_steps; } orc_stats;
int orc_kbest(int k, int N, int M, int maximize, const double *C, int use_cutoff, double cutoff,
              int32_t *col4rowBest, int32_t *row4colBest, double *gainBest, orc_stats *st);
}

static const double INF = HUGE_VAL;

struct State {  // one emitted hypothesis (kept for its lazy children)
    std::vector<double> u, v;
    std::vector<int> r4c, c4r;
    uint64_t forb[4] = {0, 0, 0, 0};
    double gain = 0;
    int activeCol = 0;
};

struct Entry { double gain; int parent; int col; };

struct Stats { long steps = 0, steps_noprune = 0, aborted = 0, children = 0, resolves = 0, visits = 0; };

struct Solver {
    int D, M;
    std::vector<double> C, spc;
    std::vector<int> pred;
    std::vector<char> scanned;
    Stats st;

    // One augmentation from column `start` on (u, v, r4c, c4r).  inScan / forbStart are
    // per-row flags.  bound: abort as soon as base + delta > bound (early termination).
    // Returns 0 ok, 1 infeasible, 2 aborted.  If `full`, applies dual update.
    int augment(std::vector<double> &u, std::vector<double> &v, std::vector<int> &r4c, std::vector<int> &c4r,
                int start, std::vector<char> inScan, const std::vector<char> *forbStart, double base, double bound,
                bool full, bool count)
    {
        std::fill(spc.begin(), spc.end(), INF);
        std::fill(scanned.begin(), scanned.end(), 0);
        std::vector<int> scannedCols;
        int sink = -1, cur = start;
        double delta = 0;
        do {
            scannedCols.push_back(cur);
            if (count) st.steps++;
            double minVal = INF;
            int closest = -1;
            for (int r = 0; r < D; r++) {
                if (!inScan[r]) continue;
                if (forbStart && cur == start && (*forbStart)[r]) continue;
                double rc = delta + C[r + (size_t)cur * D] - u[cur] - v[r];
                if (count) st.visits++;
                if (rc < spc[r]) { pred[r] = cur; spc[r] = rc; }
                if (spc[r] < minVal) { minVal = spc[r]; closest = r; }
            }
            if (minVal == INF) return 1;
            if (base + minVal > bound) return 2;
            scanned[closest] = 1;
            inScan[closest] = 0;
            delta = spc[closest];
            if (c4r[closest] == -1) sink = closest; else cur = c4r[closest];
        } while (sink == -1);
        if (full) {
            u[start] = u[start] + delta;
            for (size_t i = 1; i < scannedCols.size(); i++) {
                int c = scannedCols[i];
                u[c] = u[c] + delta - spc[r4c[c]];
            }
            for (int r = 0; r < D; r++) if (scanned[r]) v[r] = v[r] - delta + spc[r];
        }
        int r = sink, c;
        do { c = pred[r]; c4r[r] = c; int nx = r4c[c]; r4c[c] = r; r = nx; } while (c != start);
        return 0;
    }

    double gainOf(const std::vector<int> &r4c) const
    {
        double g = 0;
        for (int c = 0; c < M; c++) g = g + C[(size_t)c * D + r4c[c]];
        return g;
    }

    // child (parent P, column c): sets inScan/forb, runs the augmentation on copies.
    int child(const State &P, int c, bool full, double bound, State *out, double *gainOut, bool count)
    {
        std::vector<char> inScan(D, 0), forb(D, 0);
        for (int j = c; j < D; j++) inScan[P.r4c[j]] = 1;
        if (c == P.activeCol) { for (int r = 0; r < D; r++) forb[r] = (P.forb[r >> 6] >> (r & 63)) & 1; }
        else forb[P.r4c[c]] = 1;
        State S = P;  // copy (the device only copies what it needs)
        S.c4r[S.r4c[c]] = -1;
        S.r4c[c] = -1;
        int rc = augment(S.u, S.v, S.r4c, S.c4r, c, inScan, &forb, P.gain, bound, full, count);
        if (rc) return rc;
        S.gain = gainOf(S.r4c);
        *gainOut = S.gain;
        if (out) {
            S.activeCol = c;
            memset(S.forb, 0, sizeof(S.forb));
            for (int r = 0; r < D; r++) if (forb[r]) S.forb[r >> 6] |= 1ull << (r & 63);
            S.forb[S.r4c[c] >> 6] |= 1ull << (S.r4c[c] & 63);
            *out = S;
        }
        return 0;
    }

    int run(int k, int N, int Mc, bool maximize, const double *Cin, bool useCut, double cutoff, bool prune,
            std::vector<int> &row4col, std::vector<int> &col4row, std::vector<double> &gainBest)
    {
        D = N; M = Mc;
        C.assign((size_t)D * D, 0.0); spc.assign(D, 0); pred.assign(D, 0); scanned.assign(D, 0);
        double d = Cin[0], cmax = 0;
        for (int i = 1; i < N * M; i++) d = maximize ? std::max(d, Cin[i]) : std::min(d, Cin[i]);
        for (int i = 0; i < N * M; i++) { C[i] = maximize ? (-Cin[i] + d) : (Cin[i] - d); if (std::isfinite(C[i])) cmax = std::max(cmax, C[i]); }
        double CDelta = d * (double)M;
        std::vector<State> states;
        State root;
        root.u.assign(D, 0); root.v.assign(D, 0); root.r4c.assign(D, -1); root.c4r.assign(D, -1);
        for (int c = 0; c < D; c++) {
            std::vector<char> all(D, 1);
            if (augment(root.u, root.v, root.r4c, root.c4r, c, all, nullptr, 0, INF, true, false)) return 0;
        }
        root.gain = gainOf(root.r4c);
        root.forb[root.r4c[0] >> 6] |= 1ull << (root.r4c[0] & 63);
        auto emit = [&](const State &S, int slot) {
            for (int r = 0; r < N; r++) col4row[(size_t)slot * N + r] = S.c4r[r];
            for (int c = 0; c < M; c++) row4col[(size_t)slot * M + c] = S.r4c[c];
            gainBest[slot] = maximize ? (-S.gain + CDelta) : (S.gain + CDelta);
        };
        emit(root, 0);
        double cutoffGain = maximize ? root.gain - cutoff : root.gain + cutoff;
        states.push_back(root);
        std::vector<Entry> pool;  // sorted ascending
        int nf = k;
        for (int s = 0; s + 1 < k; s++) {
            const State &P = states[s];
            int R = k - (s + 1);
            double T = ((int)pool.size() >= R) ? pool[R - 1].gain : INF;
            double bound = INF;
            if (prune) {
                bound = T;
                if (useCut && !maximize) bound = std::min(bound, cutoffGain);
                bound = bound + 1e-9 * (std::fabs(bound) + cmax);  // safety margin
            }
            std::vector<Entry> fresh;
            for (int c = P.activeCol; c < M; c++) {
                double g;
                st.children++;
                long before = st.steps;
                int rc = child(P, c, false, bound, nullptr, &g, true);
                (void)before;
                if (rc == 2) { st.aborted++; continue; }
                if (rc == 1) continue;
                if (useCut && (maximize ? g < cutoffGain : g > cutoffGain)) continue;
                fresh.push_back({g, s, c});
            }
            for (auto &e : fresh) pool.push_back(e);
            std::stable_sort(pool.begin(), pool.end(), [](const Entry &a, const Entry &b) { return a.gain < b.gain; });
            if ((int)pool.size() > R) pool.resize(R);
            if (pool.empty()) { nf = s + 1; break; }
            Entry e = pool.front();
            pool.erase(pool.begin());
            State S;
            double g;
            st.resolves++;
            int rc = child(states[e.parent], e.col, true, INF, &S, &g, false);
            if (rc != 0 || g != e.gain) { fprintf(stderr, "re-solve mismatch rc=%d %a %a\n", rc, g, e.gain); exit(2); }
            emit(S, s + 1);
            states.push_back(S);
            if (useCut) {
                if (!maximize) { if (gainBest[s + 1] > gainBest[0] + cutoff) { nf = s + 1; break; } }
                else           { if (gainBest[s + 1] < gainBest[0] - cutoff) { nf = s + 1; break; } }
            }
        }
        return nf;
    }
};

static uint64_t sm_state;
static double u01() {
    sm_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = sm_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

int main(int argc, char **argv)
{
    int N = argc > 1 ? atoi(argv[1]) : 64, M = argc > 2 ? atoi(argv[2]) : N, k = argc > 3 ? atoi(argv[3]) : 200;
    int B = argc > 4 ? atoi(argv[4]) : 4;
    double infFrac = argc > 5 ? atof(argv[5]) : 0.0;
    double cutoff = argc > 6 ? atof(argv[6]) : -1;
    sm_state = 0x5EED0000ull + 1000 * N + k;
    long totSteps[2] = {0, 0}, totOrc = 0, totAbort = 0, totChildren = 0;
    for (int b = 0; b < B; b++) {
        std::vector<double> C((size_t)N * M);
        for (auto &x : C) { x = u01(); }
        if (infFrac > 0) for (auto &x : C) if (u01() < infFrac) x = INF;
        std::vector<int32_t> oc((size_t)k * N), orr((size_t)k * M);
        std::vector<double> og(k);
        orc_stats os;
        int onf = orc_kbest(k, N, M, 0, C.data(), cutoff >= 0, cutoff, oc.data(), orr.data(), og.data(), &os);
        totOrc += os.dijkstra_steps;
        for (int prune = 0; prune < 2; prune++) {
            Solver S;
            std::vector<int> r4c((size_t)k * M), c4r((size_t)k * N);
            std::vector<double> g(k);
            int nf = S.run(k, N, M, false, C.data(), cutoff >= 0, cutoff, prune, r4c, c4r, g);
            bool ok = nf == onf;
            for (int s = 0; ok && s < nf; s++) {
                if (memcmp(&g[s], &og[s], 8)) ok = false;
                for (int c = 0; c < M; c++) if (r4c[(size_t)s * M + c] != orr[(size_t)s * M + c]) ok = false;
                for (int r = 0; r < N; r++) if (c4r[(size_t)s * N + r] != oc[(size_t)s * N + r]) ok = false;
            }
            if (!ok) { printf("MISMATCH problem %d prune %d nf %d vs %d\n", b, prune, nf, onf); return 1; }
            totSteps[prune] += S.st.steps;
            if (prune) { totAbort += S.st.aborted; totChildren += S.st.children; }
        }
    }
    printf("N=%d M=%d k=%d B=%d: all match. oracle child steps/problem %.0f; lazy no-prune %.0f; pruned %.0f (%.1f%%), aborted children %.1f%% of %.0f\n",
           N, M, k, B, (double)totOrc / B, (double)totSteps[0] / B, (double)totSteps[1] / B,
           100.0 * totSteps[1] / totSteps[0], 100.0 * totAbort / totChildren, (double)totChildren / B);
    return 0;
}
