"""Would atoms learnt in round 1 (children of the 8 best children of the root whose own alternating path is disjoint from
their parent's) tighten the a-priori bound?  CPU experiment.  Development aid."""
import os, sys, itertools
import numpy as np
from scipy.optimize import linear_sum_assignment
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as ol
from probabilisticsemslam_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
OVERLAP = len(sys.argv) > 2
DISJ = len(sys.argv) > 3
_, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
batch = wl.dense_batch(6, N, M, seed)
BIG = 1e6

def solve(C, fixed_cols, rowof_fixed, forb):
    Cc = C.copy()
    for c2 in fixed_cols:
        Cc[:, c2] = BIG; Cc[rowof_fixed[c2], c2] = C[rowof_fixed[c2], c2]
    for (r, c) in forb: Cc[r, c] = BIG
    rr, cc = linear_sum_assignment(Cc)
    val = Cc[rr, cc].sum()
    if val >= BIG: return None, None
    rowof = np.empty(M, int); rowof[cc] = rr
    return val, rowof

def mask_of(a, b):
    m = 0
    for c in range(M):
        if a[c] != b[c]: m |= (1 << int(a[c])) | (1 << int(b[c]))
    return m

def bound(d, ms):
    order = np.argsort(d)[:64]; d = np.array(d)[order]; ms = [ms[i] for i in order]; n = len(d)
    vals = list(d) + [d[i] + d[j] for i in range(n) for j in range(i + 1, n) if ms[i] & ms[j] == 0]
    t16, t8 = min(n, 16), min(n, 8)
    vals += [d[i] + d[j] + d[l] for i, j, l in itertools.combinations(range(t16), 3) if ms[i] & ms[j] == 0 and ms[i] & ms[l] == 0 and ms[j] & ms[l] == 0]
    vals += [sum(d[a] for a in q) for q in itertools.combinations(range(t8), 4) if all(ms[a] & ms[b] == 0 for a, b in itertools.combinations(q, 2))]
    vals = np.sort(vals)
    return vals[k - 2] if len(vals) >= k - 1 else np.inf

for mi in range(6):
    C = batch[mi].reshape(M, N).T.copy()
    nf, r4c, c4r, g = ol.orc_kbest(batch[mi], N, M, k)
    gap = g[nf - 1] - g[0]
    root = r4c[0].astype(int)
    atoms = []  # (delta, mask, column, rowof)
    for c in range(M):
        val, rowof = solve(C, range(c), root, [(root[c], c)])
        if val is None: continue
        atoms.append((val - g[0], mask_of(rowof, root), c, rowof, val))
    b1 = bound([a[0] for a in atoms], [a[1] for a in atoms])
    best8 = sorted(atoms, key=lambda a: a[0])[:8]
    masks = {a[1] for a in atoms}
    extra = []
    for (dlt, m, c, rowof, val) in best8:
        forbAcc = [(root[c], c)]
        for c2 in range(c, M):   # children of this node: columns >= c; first child accumulates the forbidden arc
            forb = forbAcc + [(rowof[c2], c2)] if c2 == c else [(rowof[c2], c2)]
            v2, r2 = solve(C, range(c2), rowof, forb)
            if v2 is None: continue
            m2 = mask_of(r2, rowof)
            if m2 & m:                     # overlaps the parent's path: the grandchild against the ROOT, if that is ONE path
                mr = mask_of(r2, root)
                # connected? walk the permutation cycles of root^-1 o r2 over the moved columns
                cols = [cc for cc in range(M) if r2[cc] != root[cc]]
                colof_root = {int(root[cc]): cc for cc in range(M)}
                seen = set(); ncyc = 0
                for c0 in cols:
                    if c0 in seen: continue
                    ncyc += 1; cur = c0
                    while cur not in seen:
                        seen.add(cur); cur = colof_root[int(r2[cur])]
                if ncyc == 1 and OVERLAP:
                    masks.add(mr); extra.append((v2 - g[0], mr))
                continue
            if not DISJ: continue
            if m2 in masks: continue       # (maybe) a known atom
            masks.add(m2); extra.append((v2 - val, m2))
    b2 = bound([a[0] for a in atoms] + [e[0] for e in extra], [a[1] for a in atoms] + [e[1] for e in extra])
    print(f"{cfg} matrix {mi}: gap {gap:.4f}; bound from the root's children {b1/gap:.2f} x; with {len(extra)} atoms learnt in round 1: {b2/gap:.2f} x")
