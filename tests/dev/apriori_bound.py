"""How tight is an a-priori upper bound on the k-th best gain built from the root's children?  Every child differs from
the optimum by one alternating cycle; children with disjoint cycles combine additively into further distinct assignments
(pairs, triples).  The k-th smallest of singles + disjoint pairs + disjoint triples bounds the k-th best gain from above.
CPU experiment (scipy LAP + the oracle).  Development aid."""
import os, sys, itertools
import numpy as np
from scipy.optimize import linear_sum_assignment
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as ol
from probabilisticsemslam_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
_, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
batch = wl.dense_batch(6, N, M, seed)
BIG = 1e6
for mi in range(6):
    C = batch[mi].reshape(M, N).T.copy()   # column-major N x M -> C[r, c]
    nf, r4c, c4r, g = ol.orc_kbest(batch[mi], N, M, k)
    gap = g[nf - 1] - g[0]
    root = r4c[0]                           # row of column c
    deltas, masks = [], []
    for c in range(M):
        Cc = C.copy()
        for c2 in range(c):                 # columns before c keep their arcs
            Cc[:, c2] = BIG; Cc[root[c2], c2] = C[root[c2], c2]
        Cc[root[c], c] = BIG                # the arc of column c is forbidden
        rr, cc = linear_sum_assignment(Cc)
        val = Cc[rr, cc].sum()
        if val >= BIG: continue
        rowof = np.empty(M, int); rowof[cc] = rr
        m = 0
        for c2 in range(M):
            if rowof[c2] != root[c2]: m |= (1 << int(rowof[c2])) | (1 << int(root[c2]))
        deltas.append(val - g[0]); masks.append(m)
    order = np.argsort(deltas); d = np.array(deltas)[order]; ms = [masks[i] for i in order]
    vals = list(d)
    n = len(d)
    for i in range(n):
        for j in range(i + 1, n):
            if ms[i] & ms[j] == 0:
                vals.append(d[i] + d[j])
    top = min(n, 24)
    for i, j, l in itertools.combinations(range(top), 3):
        if ms[i] & ms[j] == 0 and ms[i] & ms[l] == 0 and ms[j] & ms[l] == 0:
            vals.append(d[i] + d[j] + d[l])
    for q in itertools.combinations(range(min(n, 16)), 4):
        ok = all(ms[a] & ms[b] == 0 for a, b in itertools.combinations(q, 2))
        if ok: vals.append(sum(d[a] for a in q))
    # variants
    v2 = list(d) + [d[i] + d[j] for i in range(n) for j in range(i + 1, n) if ms[i] & ms[j] == 0]
    b2 = np.sort(v2)[k - 2] if len(v2) >= k - 1 else np.inf
    v3 = v2 + [d[i] + d[j] + d[l] for i, j, l in itertools.combinations(range(min(n, 16)), 3) if ms[i] & ms[j] == 0 and ms[i] & ms[l] == 0 and ms[j] & ms[l] == 0]
    b3 = np.sort(v3)[k - 2]
    # subset-sum over the cheapest 12 atoms (all disjoint subsets) + pairs of everything
    v12 = list(v2)
    top12 = min(n, 12)
    for mask in range(1, 1 << top12):
        idx = [i for i in range(top12) if mask >> i & 1]
        if len(idx) < 3: continue
        if all(ms[a] & ms[b_] == 0 for a, b_ in itertools.combinations(idx, 2)): v12.append(sum(d[a] for a in idx))
    b12 = np.sort(v12)[k - 2]
    vk = list(v2)
    t16 = min(n, 16); t8 = min(n, 8)
    vk += [d[i] + d[j] + d[l] for i, j, l in itertools.combinations(range(t16), 3) if ms[i] & ms[j] == 0 and ms[i] & ms[l] == 0 and ms[j] & ms[l] == 0]
    vk += [sum(d[a] for a in q) for q in itertools.combinations(range(t8), 4) if all(ms[a] & ms[b_] == 0 for a, b_ in itertools.combinations(q, 2))]
    bk = np.sort(vk)[k - 2] if len(vk) >= k - 1 else np.inf
    print(f"   kernel recipe (pairs of all, triples of 16, quads of 8): {bk/gap:.2f} x  ({len(vk)} combos)")
    print(f"   singles+pairs {b2/gap:.2f} x; +triples(16) {b3/gap:.2f} x; +all subsets of cheapest 12 {b12/gap:.2f} x")
    vals = np.sort(np.array(vals))
    b = vals[k - 2] if len(vals) >= k - 1 else np.inf   # + the optimum itself = k assignments
    print(f"{cfg} matrix {mi}: gap {gap:.4f}; children {n}, cheapest deltas {d[:4].round(4)}; combos {len(vals)}; a-priori bound {b:.4f} = {b/gap:.2f} x gap")
