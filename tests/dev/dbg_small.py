import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol
np.set_printoptions(linewidth=200)
eng = pk.KBestEngine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
if name in wl.DENSE_CONFIGS:
    cs, N, M, k = wl.dense_config(name, B=1)
    cost = cs[0]; cut = None
else:
    fr = wl.kitti_like_frames(1)[0]
    cost, idx = ol.condition_costs(fr, 30, 10); N = len(idx); M = 10; k = int(sys.argv[2]) if len(sys.argv) > 2 else 20; cut = 42.0
nf, r4c, c4r, g = eng.kbest(cost.reshape(1, -1), N, M, k, False, cut)
onf, or4c, oc4r, og = ol.orc_kbest(cost, N, M, k, False, cut)
print("nf", nf[0], onf)
for s in range(min(int(nf[0]), onf, 12)):
    print(s, "gpu", g[0][s], r4c[0][s], "| orc", og[s], or4c[s], "OK" if (r4c[0][s] == or4c[s]).all() and g[0][s] == og[s] else "DIFF")
