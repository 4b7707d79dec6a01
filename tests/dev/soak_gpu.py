"""Randomised soak: HIP engine vs the CPU oracle over many shapes / modes (development aid, run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probabilisticsemslam_amd as pk
import oracle_lib as ol

def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = pk.KBestEngine(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
for t in range(trials):
    N = int(rng.integers(1, 65)); M = int(rng.integers(1, N + 1)); k = int(rng.integers(1, 260))
    B = int(rng.choice([1, 2, 3, 5, 9, 300, 600])) if N <= 24 else int(rng.integers(1, 7))
    scale = float(rng.choice([1.0, 20.0, 1e-3, 1e4]))
    costs = rng.random((B, N * M)) * scale - (0.3 * scale if t % 3 == 0 else 0.0)
    mode = t % 5
    if mode == 1: costs[rng.random((B, N * M)) < 0.35] = np.inf
    if mode == 4: costs = np.round(costs / scale * 6) * scale  # many exact ties
    maximize = mode == 2
    cutoff = [None, None, None, 0.15 * scale, None][mode]
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k, maximize, cutoff)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, maximize, cutoff)
    ok = (nf == onf).all()
    for b in range(B):
        n = min(nf[b], onf[b])
        if mode == 4:
            # equal-gain assignments may come out in another order, and with an inexact scale their serial sums can
            # differ in the last bit: compare the sorted gains to 1e-12
            ok = ok and np.allclose(np.sort(g[b, :n]), np.sort(og[b, :n]), rtol=1e-12, atol=1e-12 * scale)
        else:
            ok = ok and (r4c[b, :n] == or4c[b, :n]).all() and (c4r[b, :n] == oc4r[b, :n]).all() and (bits(g[b, :n]) == bits(og[b, :n])).all()
    if not ok:
        bad += 1
        print("MISMATCH trial", t, N, M, k, B, mode, "scale", scale, "nf equal", (nf == onf).all(), flush=True)
        for b in range(B):
            n = min(nf[b], onf[b])
            if nf[b] != onf[b] or sorted(g[b, :n].tolist()) != sorted(og[b, :n].tolist()):
                d = np.array(sorted(g[b, :n].tolist())) - np.array(sorted(og[b, :n].tolist()))
                print("  problem", b, "nf", nf[b], onf[b], "max |diff| of sorted gains", np.abs(d).max() if n else 0, flush=True)
                break
print("soak trials", trials, "bad =", bad)
