"""Host model: a bounded depth-first walk (partial sums <= U, columns in the reference's order) finds every assignment with gain
<= U; with U raised until k of them exist it finds the k best.  For KITTI-like frames: the checker's k-th best gain, and per pass
of the bound search (U = greedy gain, then doubled until k leaves) the number of partial assignments visited and of leaves.
python3 tests/dev/proto_bnb.py [frames] [nL] [nM] [dense]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from probabilisticsemslam_amd import workloads as wl

F = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nL = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nM = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dense = len(sys.argv) > 4
k = 200
frames = wl.kitti_like_frames(F, nL=nL, nM=nM)
if dense:
    rng = np.random.default_rng(1)
    frames = []
    for _ in range(F):
        C = np.full((nL + nM) * nM, np.inf)
        for c in range(nM):
            C[c * (nL + nM): c * (nL + nM) + nL] = rng.random(nL) * 30.0
            C[c * (nL + nM) + nL + c] = 10.0
        frames.append(C)
tot_nodes = []
for f in frames:
    cond, idx = ol.condition_costs(f, nL + nM, nM)
    N, M = len(idx), nM
    C = cond.reshape(M, N)
    nf, r4c, c4r, g = ol.orc_kbest(cond, N, M, k, cutoff=42.0)
    gk = g[nf - 1]
    used, gsum = set(), 0.0
    for c in range(M):
        r = min((C[c][r], r) for r in range(N) if r not in used)[1]
        used.add(r); gsum += C[c][r]
    Umax = gsum + 42.0
    def run(U, budget=2_000_000):
        cnt = [0, 0]
        def walk(c, acc, usedm):
            if c == M:
                cnt[1] += 1
                return
            col = C[c]
            for r in range(N):
                if (usedm >> r) & 1: continue
                a = acc + col[r]
                if a > U: continue
                cnt[0] += 1
                if cnt[0] > budget: return
                walk(c + 1, a, usedm | (1 << r))
        walk(0, 0.0, 0)
        return cnt
    U = max(gsum, 1e-3)
    log = []
    total = 0
    while True:
        nodes, leaves = run(U)
        total += nodes
        log.append(f"U={U:.2f}: {nodes} nodes / {leaves} leaves")
        if leaves >= k or U >= Umax: break
        U = min(Umax, 2 * U)
    # final pass at the k-th gain itself (what the histogram's bucket edge gives)
    nodes, leaves = run(gk * (1 + 1e-9))
    total += nodes
    tot_nodes.append(total)
    print(f"N={N} M={M} best {g[0]:.2f} k-th {gk:.2f} greedy {gsum:.2f} | " + "; ".join(log) + f" | final at k-th: {nodes} nodes / {leaves} leaves | all passes {total} nodes")
print("mean nodes per frame over all passes:", int(np.mean(tot_nodes)), "max", max(tot_nodes))
