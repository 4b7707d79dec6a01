"""dev aid: where does the general-size kernel differ from the checker beyond 512 rows?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import probabilisticsemslam_amd as pk

def run(N, M, k, env):
    for kk, v in env.items():
        os.environ[kk] = v
    eng = pk.KBestEngine(0)
    for kk in env:
        del os.environ[kk]
    rng = np.random.default_rng(N + M)
    costs = rng.random((1, N * M))
    kw = {}
    if "NO_PRUNE" in env.get("TAG", ""):
        kw["prune"] = False
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k, reorder="NOREORDER" not in env.get("TAG", ""), **kw)
    onf, or4c, oc4r, og = ol.orc_kbest(costs[0], N, M, k)
    bad = [s for s in range(min(int(nf[0]), onf)) if not (r4c[0, s] == or4c[s]).all()]
    gb = [s for s in range(min(int(nf[0]), onf)) if g[0, s] != og[s]]
    print(f"{N}x{M} k={k} {env}: nf {nf[0]}/{onf}, slots with other row4col {bad[:6]}, other gain {gb[:6]}",
          [(float(g[0, s]), float(og[s])) for s in gb[:2]], flush=True)
    eng.close()

for (N, M, k) in ((1024, 1024, 10), (1024, 512, 10), (768, 768, 10), (1000, 1000, 10), (1024, 1024, 3), (640, 640, 10)):
    run(N, M, k, {})
run(1024, 1024, 10, {"TAG": "NOREORDER"})
run(1024, 1024, 10, {"KBEST_NO_T0": "1"})
run(1024, 1024, 10, {"TAG": "NO_PRUNE"})
