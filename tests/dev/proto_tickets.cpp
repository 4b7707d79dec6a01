// proto_tickets.cpp -- DEV TOOL (derived from proto_colorder.cpp) (host only, no GPU, no checker): counts the work of the device algorithm of
// csrc/kbest_engine.hip -- batched frontier, first-step filter with the backward bound, early termination, bounded pool --
// under different COLUMN ORDERS of Murty's partition (split, shortestPathCPP.cpp:455-532):
//   0  the reference's order (columns as they come)
//   1  one static order per problem: columns by the exact cost of taking their row away at the root, dear first
//      (what the kernels do since round 3)
//   3  as 2 with the exact key (a full search per free column and node: what a per-node order could give at best)
//   2  per node: the active column first (its accumulated exclusions stay a single-column affair), the other free columns
//      re-sorted at every split by a key computed from the node's own duals (Miller-Stone-Cox), dear first
// Prints per problem: children that pass the filter, Dijkstra steps, row visits, completed children, rounds.
// Build: g++ -O2 -std=c++17 -o /tmp/proto_colorder tests/dev/proto_colorder.cpp
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static const double INF = HUGE_VAL;
typedef unsigned long long u64;

struct Node {
    std::vector<double> u, v;
    std::vector<int> r4c, c4r;
    double gain = 0;
    u64 fixed = 0;      // columns whose row is fixed in every assignment of this node's set
    int activeCol = 0;  // the column this node was made on (its exclusions accumulate)
    u64 forb = 0;       // rows excluded for activeCol
};

struct Counters { long started = 0, filtered = 0, steps = 0, visits = 0, completed = 0, rounds = 0, keyPasses = 0; };

struct Solver {
    int D;
    std::vector<double> C;  // column-major D x D
    Counters ct;
    int mode;
    std::vector<int> order;  // static order (modes 0, 1): position -> column
    double lastDelta = 0, lastLB = 0;
    double rho = 1.0, kappa = 0.25, rho1 = 1.0, phi = 1e9, rhoNow = 1.0, adaptA = 0.0, adaptB = 0.0, rhoMin = 0.0;
    int minPool = 8;
    double Tcap = INF;       // an upper bound of the k-th best gain known from the start (the cutoff variant fed the answer)
    int beamW = 0, beamE = 6;  // beam width / rows tried per entry and column (0: no beam)
    long beamWork = 0;
    double beamGap = 0;      // the beam's bound on (k-th best gain - optimum)
    double lastGain = 0;     // gain of the last hypothesis emitted
    double rootGain = 0;

    double rc(const Node &P, int r, int c) const { return (C[r + (size_t)c * D] - P.u[c]) - P.v[r]; }

    // Dijkstra of one child: start column c, candidate rows `cand` (bit mask), rows skipped while scanning the start column
    // `forbStart`; the only sink is row fr.  Returns 0 = completed, 2 = abandoned, 1 = infeasible.
    int search(const Node &P, int c, u64 cand, u64 forbStart, int fr, double bound, double minIn, Node *out)
    {
        std::vector<double> spc(D, INF);
        std::vector<int> pred(D, -1);
        u64 left = cand, scanned = 0;
        int cur = c;
        double delta = 0;
        const double tight = bound - minIn;
        bool useTight = minIn > 0;
        int sink = -1;
        for (;;) {
            ct.steps++;
            double mn = INF;
            int arg = -1;
            for (int r = 0; r < D; r++) {
                if (!((left >> r) & 1ull)) continue;
                if (cur == c && ((forbStart >> r) & 1ull)) { if (spc[r] < mn) { mn = spc[r]; arg = r; } continue; }
                ct.visits++;
                const double t = ((delta + C[r + (size_t)cur * D]) - P.u[cur]) - P.v[r];
                if (t < spc[r]) { spc[r] = t; pred[r] = cur; }
                if (spc[r] < mn) { mn = spc[r]; arg = r; }
            }
            if (!(mn < INF)) return 1;
            lastLB = mn;
            if (mn > bound) return 2;
            if (useTight && mn > tight) {
                if (spc[fr] > bound) return 2;
                useTight = false;
            }
            delta = mn;
            left &= ~(1ull << arg);
            scanned |= 1ull << arg;
            if (arg == fr) { sink = arg; lastDelta = delta; break; }
            cur = P.c4r[arg];
        }
        if (out) {
            *out = P;
            Node &S = *out;
            S.c4r[fr] = -1;
            S.r4c[c] = -1;
            for (int r = 0; r < D; r++)
                if ((scanned >> r) & 1ull) {
                    if (r != sink) { const int cc = P.c4r[r]; S.u[cc] = P.u[cc] + delta - spc[r]; }
                    S.v[r] = P.v[r] - delta + spc[r];
                }
            S.u[c] = P.u[c] + delta;
            int r = sink, cc;
            do { cc = pred[r]; S.c4r[r] = cc; const int nx = S.r4c[cc]; S.r4c[cc] = r; r = nx; } while (cc != c);
            double g = 0;
            for (int j = 0; j < D; j++) g += C[S.r4c[j] + (size_t)j * D];
            S.gain = g;
        }
        return 0;
    }

    struct Entry { double gain; Node node; bool split; };

    void run(int k, int spec, int N, const double *Cin)
    {
        D = N;
        C.assign(Cin, Cin + (size_t)D * D);
        double mn = INF;
        for (double x : C) mn = std::min(mn, x);
        double cmax = 0;
        for (double &x : C) { x -= mn; cmax = std::max(cmax, x); }
        // root: plain successive shortest paths
        Node root;
        root.u.assign(D, 0); root.v.assign(D, 0); root.r4c.assign(D, -1); root.c4r.assign(D, -1);
        for (int c = 0; c < D; c++) {
            std::vector<double> spc(D, INF);
            std::vector<int> pred(D, -1);
            u64 left = (D >= 64) ? ~0ull : ((1ull << D) - 1), scanned = 0;
            int cur = c, sink = -1;
            double delta = 0;
            std::vector<int> cols;
            while (sink < 0) {
                cols.push_back(cur);
                double m = INF; int arg = -1;
                for (int r = 0; r < D; r++) {
                    if (!((left >> r) & 1ull)) continue;
                    const double t = ((delta + C[r + (size_t)cur * D]) - root.u[cur]) - root.v[r];
                    if (t < spc[r]) { spc[r] = t; pred[r] = cur; }
                    if (spc[r] < m) { m = spc[r]; arg = r; }
                }
                delta = m; left &= ~(1ull << arg); scanned |= 1ull << arg;
                if (root.c4r[arg] < 0) sink = arg; else cur = root.c4r[arg];
            }
            root.u[c] += delta;
            for (size_t i = 1; i < cols.size(); i++) root.u[cols[i]] += delta - spc[root.r4c[cols[i]]];
            for (int r = 0; r < D; r++) if ((scanned >> r) & 1ull) root.v[r] += spc[r] - delta;
            int r = sink, cc;
            do { cc = pred[r]; root.c4r[r] = cc; const int nx = root.r4c[cc]; root.r4c[cc] = r; r = nx; } while (cc != c);
        }
        root.gain = 0;
        for (int j = 0; j < D; j++) root.gain += C[root.r4c[j] + (size_t)j * D];
        // static order
        order.resize(D);
        for (int i = 0; i < D; i++) order[i] = i;
        if (mode >= 1) {
            std::vector<double> key(D);
            Counters save = ct;
            for (int c = 0; c < D; c++) {
                Node tmp;
                const int fr = root.r4c[c];
                const u64 all = (D >= 64) ? ~0ull : ((1ull << D) - 1);
                Node P = root;
                P.c4r[fr] = -1;
                // exact cost of taking column c's row away
                std::vector<double> spc(D, INF);
                u64 left = all;
                int cur = c;
                double delta = 0;
                for (;;) {
                    double m = INF; int arg = -1;
                    for (int r = 0; r < D; r++) {
                        if (!((left >> r) & 1ull)) continue;
                        if (!(cur == c && r == fr)) {
                            const double t = ((delta + C[r + (size_t)cur * D]) - root.u[cur]) - root.v[r];
                            if (t < spc[r]) spc[r] = t;
                        }
                        if (spc[r] < m) { m = spc[r]; arg = r; }
                    }
                    delta = m;
                    if (!(m < INF) || arg == fr) break;
                    left &= ~(1ull << arg);
                    cur = root.c4r[arg];
                }
                key[c] = delta;
            }
            ct = save;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b]; });
        }
        rootGain = root.gain;
        if (beamW > 0) {
            // A BEAM in reduced-cost space as a bound generator: the gain of any assignment is the optimum plus the
            // sum of its reduced costs (C - u - v >= 0 at the root), so a walk over the columns that keeps only the beamW best partial
            // assignments per level -- by partial sum + a lower bound of what the columns to come must add (0 where a column's optimal
            // row is still free, its cheapest other free row's reduced cost where not) -- ends with <= beamW DISTINCT complete
            // assignments: the k-th smallest of their sums bounds the k-th best gain from above, whatever the beam threw away.
            struct BE { u64 used; double sum, pri; };
            std::vector<BE> cur{{0ull, 0.0, 0.0}}, nxt;
            std::vector<std::vector<std::pair<double, int>>> byCol(D);
            for (int c = 0; c < D; c++) {
                for (int r = 0; r < D; r++) byCol[c].push_back({std::max(0.0, rc(root, r, c)), r});
                std::sort(byCol[c].begin(), byCol[c].end());
            }
            for (int lvl = 0; lvl < D; lvl++) {
                const int c = order[lvl];
                nxt.clear();
                for (const BE &e : cur) {
                    int taken = 0;
                    for (auto &pr : byCol[c]) {
                        if ((e.used >> pr.second) & 1ull) continue;
                        if (taken++ >= beamE) break;
                        beamWork++;
                        BE n{e.used | (1ull << pr.second), e.sum + pr.first, 0.0};
                        double lb = n.sum;
                        for (int l2 = lvl + 1; l2 < D; l2++) {  // columns to come: optimal row free -> 0, else the cheapest free row
                            const int c2 = order[l2];
                            if (!((n.used >> root.r4c[c2]) & 1ull)) continue;
                            double m = INF;
                            for (auto &q : byCol[c2]) if (!((n.used >> q.second) & 1ull)) { m = q.first; break; }
                            lb += m;
                        }
                        n.pri = lb;
                        if (lb < INF) nxt.push_back(n);
                    }
                }
                std::sort(nxt.begin(), nxt.end(), [](const BE &a, const BE &b) { return a.pri < b.pri; });
                if ((int)nxt.size() > beamW) nxt.resize(beamW);
                cur.swap(nxt);
            }
            std::vector<double> leaves;
            for (auto &e : cur) leaves.push_back(e.sum);
            std::sort(leaves.begin(), leaves.end());
            if ((int)leaves.size() >= k) { beamGap = leaves[k - 1]; if (root.gain + beamGap < Tcap) Tcap = root.gain + beamGap * (1.0 + 1e-9) + 1e-12; }
        }
        root.activeCol = order[0];
        root.forb = 1ull << root.r4c[order[0]];
        root.fixed = 0;
        std::vector<int> posOf(D);
        for (int i = 0; i < D; i++) posOf[order[i]] = i;

        // ---- optimistic bounds with re-split tickets ------------------------------------------------------------------
        // Every node is split against an OPTIMISTIC bound Tg <= the pool's valid threshold T (the rho-quantile of the pool's
        // candidates).  Children that die between the two are not lost: the smallest lower bound among them (LB_P) goes into
        // the pool as a TICKET of the node; a ticket at the pool's head blocks emission and is processed like a candidate:
        // the node is split again against a higher bound (children that completed before are skipped: done mask).
        struct NodeX { Node n; u64 done = 0; };
        struct PE { double key; int id; bool split; bool ticket; };
        std::vector<NodeX> nodes;
        nodes.push_back({root, 0});
        std::vector<PE> pool;
        int emitted = 1;
        struct Sel { int id; bool ticket; double key; };
        std::vector<Sel> sel{{0, false, root.gain}};
        const double opt = root.gain;
        while (emitted < k && !sel.empty()) {
            ct.rounds++;
            const int R = k - emitted;
            if (ct.rounds == 1) rhoNow = rho;
            std::vector<double> cg;
            for (auto &e : pool) if (!e.ticket) cg.push_back(e.key);
            double T = ((int)cg.size() >= R) ? cg[R - 1] : INF;
            if (ct.rounds >= 2 && Tcap < T) T = Tcap;  // a valid a-priori threshold (the kernel's T0 / T1), from round 1 on
            double Tg = T;
            if (rho < 1.0 && (int)cg.size() >= minPool) {
                const int n = std::min((int)cg.size(), R);
                const double fr = std::min(1.0, (double)emitted / (phi * k));
                const double rhoE = adaptA > 0 ? rhoNow : rho + (rho1 - rho) * fr;  // the quantile grows with the share of the answer that is out
                int qi = (int)(rhoE * n);
                if (qi >= n) qi = n - 1;
                if (cg[qi] < Tg) Tg = cg[qi];
            }
            std::vector<PE> fresh;
            int newTickets = 0;
            for (const Sel &s : sel) {
                const Node P = nodes[s.id].n;
                double bAbs = Tg;
                if (s.ticket) { const double up = s.key + kappa * (s.key - opt) + 1e-12; if (up > bAbs) bAbs = up; ct.keyPasses++; }
                if (bAbs > T) bAbs = T;
                const double bound = (bAbs < INF) ? (bAbs - P.gain) + 1e-9 * (std::fabs(bAbs) + cmax) : INF;
                const double boundV = (T < INF) ? (T - P.gain) + 1e-9 * (std::fabs(T) + cmax) : INF;
                double lbP = INF;
                std::vector<int> seq;
                for (int i = posOf[P.activeCol]; i < D; i++) seq.push_back(order[i]);
                u64 fixedSoFar = P.fixed;
                for (size_t i = 0; i < seq.size(); i++) {
                    const int c = seq[i];
                    const int fr = P.r4c[c];
                    if (!((nodes[s.id].done >> c) & 1ull)) {
                        u64 cand = 0;
                        for (size_t j = i; j < seq.size(); j++) cand |= 1ull << P.r4c[seq[j]];
                        const u64 forbStart = (c == P.activeCol) ? P.forb : (1ull << fr);
                        double m = INF, minIn = INF;
                        for (int r = 0; r < D; r++)
                            if (((cand >> r) & 1ull) && !((forbStart >> r) & 1ull)) m = std::min(m, rc(P, r, c));
                        for (size_t j = i + 1; j < seq.size(); j++) minIn = std::min(minIn, std::max(0.0, rc(P, fr, seq[j])));
                        ct.filtered++;
                        const double lbF = m + (minIn < INF ? minIn : INF);
                        if (m < INF && !(lbF > bound)) {
                            ct.started++;
                            Node S;
                            const int st = search(P, c, cand, forbStart, fr, bound, (bound < INF && minIn < INF) ? minIn : 0.0, &S);
                            if (st == 0) {
                                ct.completed++;
                                S.fixed = fixedSoFar;
                                S.activeCol = c;
                                S.forb = forbStart | (1ull << S.r4c[c]);
                                nodes[s.id].done |= 1ull << c;
                                nodes.push_back({S, 0});
                                fresh.push_back({S.gain, (int)nodes.size() - 1, false, false});
                            } else if (st == 2) {
                                const double lb = std::max(lastLB, bound);  // the distance reached when it was given up
                                if (!(lb > boundV)) lbP = std::min(lbP, P.gain + lb);
                            }
                        } else if (m < INF && !(lbF > boundV)) lbP = std::min(lbP, P.gain + lbF);
                    }
                    fixedSoFar |= 1ull << c;
                }
                if (lbP < INF) fresh.push_back({lbP, s.id, false, true});
                if (s.ticket) newTickets++;  // (the feedback counts the re-splits, not the tickets written: most of those are never due)
            }
            if (adaptA > 0) {  // feedback: every ticket pushes the quantile towards the valid threshold, quiet rounds let it sink back
                rhoNow = std::min(1.0, rhoNow + adaptA * newTickets);
                if (!newTickets) rhoNow = std::max(rhoMin, rhoNow - adaptB);
            }
            for (auto &e : fresh) pool.push_back(e);
            std::stable_sort(pool.begin(), pool.end(), [](const PE &a, const PE &b) { return a.key < b.key || (a.key == b.key && a.ticket && !b.ticket); });
            {   // keep the R smallest candidates, and the tickets below the R-th of them
                int nc = 0;
                double Tn = INF;
                std::vector<PE> kept;
                for (auto &e : pool) {
                    if (e.ticket) { if (nc < R) kept.push_back(e); continue; }
                    if (nc < R) { kept.push_back(e); nc++; if (nc == R) Tn = e.key; }
                }
                (void)Tn;
                pool.swap(kept);
            }
            size_t h = 0;
            while (h < pool.size() && !pool[h].ticket && pool[h].split && emitted < k) { emitted++; lastGain = pool[h].key; h++; }
            pool.erase(pool.begin(), pool.begin() + h);
            sel.clear();
            {
                std::vector<PE> rest;
                bool first = true;
                for (auto &e : pool) {
                    if ((int)sel.size() < spec && (e.ticket || !e.split)) {
                        sel.push_back({e.id, e.ticket, e.key});
                        if (e.ticket) { first = false; continue; }  // a ticket leaves the pool while its node is split again
                        e.split = true;
                        if (first && emitted < k) { emitted++; lastGain = e.key; first = false; continue; }  // the head: emitted now
                    }
                    first = false;
                    rest.push_back(e);
                }
                pool.swap(rest);
            }
            if (pool.empty() && sel.empty()) break;
        }
    }
};

static u64 sm_state;
static double u01()
{
    sm_state += 0x9E3779B97F4A7C15ull;
    u64 z = sm_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 64, k = argc > 2 ? atoi(argv[2]) : 200, B = argc > 3 ? atoi(argv[3]) : 8;
    const int spec = argc > 4 ? atoi(argv[4]) : 12;
    const double t0slack = getenv("T0SLACK") ? atof(getenv("T0SLACK")) : 0.0;  // emulate the kernel's a-priori threshold: optimum + t0slack * gap
    for (int cfg = 5; cfg < argc || cfg == 5; cfg += 3) {
        const double rho = argc > cfg ? atof(argv[cfg]) : 1.0, kappa = argc > cfg + 1 ? atof(argv[cfg + 1]) : 0.25;
        const int minPool = argc > cfg + 2 ? atoi(argv[cfg + 2]) : 8;
        sm_state = 0x5EED0000ull + 1000 * N + k;
        Counters tot;
        double chk = 0, beamRatio = 0;
        long beamWorkTot = 0;
        for (int b = 0; b < B; b++) {
            std::vector<double> C((size_t)N * N);
            for (auto &x : C) x = u01();
            Solver S;
            S.mode = 1;
            S.rho = rho; S.kappa = kappa; S.minPool = minPool;
            if (getenv("RHO1")) S.rho1 = atof(getenv("RHO1")); else S.rho1 = rho;
            if (getenv("PHI")) S.phi = atof(getenv("PHI"));
            if (getenv("ADAPT_A")) S.adaptA = atof(getenv("ADAPT_A"));
            if (getenv("ADAPT_B")) S.adaptB = atof(getenv("ADAPT_B"));
            S.rhoMin = getenv("RHO_MIN") ? atof(getenv("RHO_MIN")) : rho;
            if (t0slack > 0) { Solver S0; S0.mode = 1; S0.run(k, spec, N, C.data()); S.Tcap = S0.rootGain + t0slack * (S0.lastGain - S0.rootGain); }
            if (getenv("BEAM")) S.beamW = atoi(getenv("BEAM"));
            if (getenv("BEAM_E")) S.beamE = atoi(getenv("BEAM_E"));
            S.run(k, spec, N, C.data());
            beamRatio += S.beamGap / (S.lastGain - S.rootGain);
            beamWorkTot += S.beamWork;
            chk += S.lastGain;
            tot.started += S.ct.started; tot.filtered += S.ct.filtered; tot.steps += S.ct.steps; tot.visits += S.ct.visits;
            tot.completed += S.ct.completed; tot.rounds += S.ct.rounds; tot.keyPasses += S.ct.keyPasses;
        }
        printf("N=%d k=%d spec=%d rho %.2f kappa %.2f minPool %d: children filtered %.0f, pass %.0f, steps %.0f, row visits %.0f, completed %.0f, rounds %.1f, tickets %.1f  [sum of k-th gains %.12f]\n",
               N, k, spec, rho, kappa, minPool, (double)tot.filtered / B, (double)tot.started / B, (double)tot.steps / B, (double)tot.visits / B,
               (double)tot.completed / B, (double)tot.rounds / B, (double)tot.keyPasses / B, chk);
        if (getenv("BEAM")) printf("   beam %s x %s rows: bound / true gap %.3f, (entry, row) expansions per problem %.0f\n", getenv("BEAM"), getenv("BEAM_E") ? getenv("BEAM_E") : "6", beamRatio / B, (double)beamWorkTot / B);
    }
    return 0;
}
