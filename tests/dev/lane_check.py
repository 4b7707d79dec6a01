"""Lane-per-child kernel (kbest_lane.hip) against the oracle, then kernel times of C2 / C3 for launch shapes.
Development aid (GPU box):  python tests/dev/lane_check.py [check|time|all]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol

mode = sys.argv[1] if len(sys.argv) > 1 else "all"


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env:
        del os.environ[k_]
    return e


def canon(c4r, M):
    c = c4r.copy(); c[c >= M] = -1; return c


def check():
    rng = np.random.default_rng(11)
    bad = 0
    cases = [(16, 16, 50, 40, {}), (32, 32, 200, 12, {}), (8, 8, 10, 30, {}), (20, 20, 64, 16, {}), (32, 32, 7, 20, {}),
             (24, 10, 100, 20, {}), (16, 5, 30, 20, {}), (30, 30, 200, 8, {"maximize": True}), (12, 12, 300, 10, {}),
             (32, 20, 150, 10, {"cutoff": 0.3}), (16, 16, 50, 16, {"cutoff": 0.05}), (5, 5, 200, 10, {}), (1, 1, 3, 4, {}), (3, 2, 9, 6, {})]
    for nw, spec, lg in ((1, 1, 4), (2, 3, 4), (4, 6, 4), (1, 8, 4), (2, 6, 2), (1, 3, 2), (4, 8, 2)):
        if True:
            eng = engine(KBEST_FORCE_LANE=1, KBEST_LANE_NW=nw, KBEST_LANE_SPEC=spec, KBEST_LANE_G=lg)
            for (N, M, k, B, kw) in cases:
                costs = rng.random((B, N * M))
                if N == 12: costs = np.floor(costs * 4)  # ties
                t0 = time.time()
                nf, r4c, c4r, g = eng.kbest(costs, N, M, k, **kw)
                onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, **kw)
                ok = (nf == onf).all()
                if N == 12:  # exact ties: multisets of gains
                    ok = ok and all((np.sort(g[b, :nf[b]]) == np.sort(og[b, :onf[b]])).all() for b in range(B))
                else:
                    for b in range(B):
                        n = nf[b]
                        ok = ok and (r4c[b, :n] == or4c[b, :n]).all() and (g[b, :n].view(np.int64) == og[b, :n].view(np.int64)).all()
                        ok = ok and (canon(c4r[b, :n], M) == canon(oc4r[b, :n], M)).all()
                if not ok:
                    bad += 1
                    b = 0
                    print(f"MISMATCH nw={nw} spec={spec} {N}x{M} k={k} {kw}: nf {nf[:6]} vs {onf[:6]}; g0 {g[0,:4]} vs {og[0,:4]}")
            eng.close()
    print("lane check:", "OK" if bad == 0 else f"{bad} mismatching cases")
    return bad


def time_cfg(name, B, N, k, envs):
    costs = torch.from_numpy(wl.dense_batch(B, N, N, 0x5EED0000 + 1000 * N + k)).to(dev)
    r4c = torch.empty((B, k, N), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
    ref = None
    for env in envs:
        eng = engine(**env)
        eng.reserve(B, N, k)
        eng.kbest_dev(costs, B, N, N, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.kbest_dev(costs, B, N, N, k, r4c, c4r, g, nf, stream=s)
        e1.record(); torch.cuda.synchronize()
        gs = g.sum().item(); nfs = int(nf.sum().item())
        if ref is None: ref = (gs, nfs)
        print(f"{name} B={B} {N}x{N} k={k} {env}: {e0.elapsed_time(e1)/5:.3f} ms  nf {nfs} gsum {gs:.9e} {'' if (gs, nfs) == ref else '  <-- differs from the first'}", flush=True)
        eng.close()


if mode in ("check", "all"):
    check()
if mode in ("time", "all"):
    base = [{"KBEST_NO_LANE": 1}]
    grid16 = [{"KBEST_FORCE_LANE": 1, "KBEST_LANE_NW": nw, "KBEST_LANE_SPEC": sp, "KBEST_LANE_G": lg} for lg in (4,) for nw in (1, 2) for sp in (3, 4, 6, 8)]
    grid32 = [{"KBEST_FORCE_LANE": 1, "KBEST_LANE_NW": nw, "KBEST_LANE_SPEC": sp, "KBEST_LANE_G": lg} for lg in (4,) for nw in (1, 2) for sp in (3, 4, 6, 8)]
    time_cfg("c2", 1024, 16, 50, base + grid16)
    time_cfg("c3", 4096, 32, 200, base + grid32)
