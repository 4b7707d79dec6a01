import sys, time, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import probabilisticsemslam_amd as pk
import oracle_lib as ol
eng = pk.KBestEngine(0)
rng = np.random.default_rng(3)
B, N, M, k = 120000, 6, 6, 8
costs = rng.random((B, N * M)) * 9
t0 = time.perf_counter()
nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
print("huge batch: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
idx = rng.integers(0, B, 300)
onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs[idx], N, M, k)
ok = (nf[idx] == onf).all() and (r4c[idx] == or4c).all() and (g[idx].view(np.int64) == og.view(np.int64)).all()
print("sample parity", ok)
