import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol
np.set_printoptions(linewidth=220)
eng = pk.KBestEngine(0)
cs, N, M, k = wl.dense_config("c3", B=2)
for kk in (2, 3, 5, 10, 40, 200):
    nf, r4c, c4r, g = eng.kbest(cs[1:2], N, M, kk)
    onf, or4c, oc4r, og = ol.orc_kbest(cs[1], N, M, kk)
    bad = [s for s in range(min(int(nf[0]), onf)) if not ((r4c[0][s] == or4c[s]).all() and g[0][s] == og[s])]
    print("k", kk, "nf", nf[0], onf, "first bad", bad[:5])
    if bad:
        s = bad[0]
        print(" gpu", g[0][s], r4c[0][s]); print(" orc", og[s], or4c[s])
        valid = len(set(r4c[0][s].tolist())) == M
        print(" valid perm:", valid, " gain recomputed:", sum(cs[1][c * N + r4c[0][s][c]] for c in range(M)))
        break
