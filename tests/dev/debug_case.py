import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import probabilisticsemslam_amd as pk
eng = pk.KBestEngine(0)
rng = np.random.default_rng(2024)
bad = 0
for trial in range(60):
    N = int(rng.integers(1, 65)); M = int(rng.integers(1, N + 1)); k = int(rng.integers(1, 80)); B = int(rng.integers(1, 6))
    costs = rng.random((B, N * M)) * 20 - 5
    mode = trial % 4
    if mode == 1:
        costs[rng.random((B, N * M)) < 0.4] = np.inf
    maximize = mode == 2
    cutoff = [None, None, None, 4.0][mode]
    for prune in (True, False):
        nf, r4c, c4r, g = eng.kbest(costs, N, M, k, maximize, cutoff, prune=prune)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, maximize, cutoff)
        for b in range(B):
            n = min(nf[b], onf[b])
            ok = nf[b] == onf[b] and (r4c[b, :n] == or4c[b, :n]).all() and (g[b, :n].view(np.int64) == og[b, :n].view(np.int64)).all() and (c4r[b,:n] == oc4r[b,:n]).all()
            if not ok:
                bad += 1
                first = next((s for s in range(n) if not ((r4c[b, s] == or4c[b, s]).all() and g[b, s] == og[b, s])), None)
                print(f"trial {trial} prune={prune} b={b} N={N} M={M} k={k} mode={mode} nf={nf[b]} onf={onf[b]} first_bad_slot={first}")
                if first is not None and bad < 4:
                    print("  gpu g", g[b, max(0,first-1):first+2], "orc g", og[b, max(0,first-1):first+2])
                    print("  gpu r4c", r4c[b, first], "\n  orc r4c", or4c[b, first])
print("bad", bad)
