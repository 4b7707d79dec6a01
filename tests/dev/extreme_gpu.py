import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) if '__file__' in globals() else '/root/repo'
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import probabilisticsemslam_amd as pk
import oracle_lib as ol
eng = pk.KBestEngine(0)
rng = np.random.default_rng(11)
bad = 0
for t in range(120):
    N = int(rng.integers(2, 65)); M = int(rng.integers(1, N + 1)); k = int(rng.integers(1, 220)); B = int(rng.integers(1, 4))
    scale = float(rng.choice([1e150, 1e-150, 1e10, 1e-10, 1e80]))
    costs = (rng.random((B, N * M)) - 0.5 * (t % 2)) * scale
    if t % 5 == 0: costs += 1e6 * scale          # large common offset: tiny relative differences
    if t % 7 == 0: costs[rng.random((B, N * M)) < 0.3] = np.inf
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k, maximize=(t % 3 == 0))
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, t % 3 == 0)
    ok = (nf == onf).all()
    for b in range(B):
        n = min(nf[b], onf[b])
        ok = ok and (r4c[b, :n] == or4c[b, :n]).all() and (g[b, :n].view(np.int64) == og[b, :n].view(np.int64)).all()
    if not ok:
        bad += 1; print("MISMATCH", t, N, M, k, B, scale, flush=True)
print("extreme-scale trials 120 bad =", bad)
