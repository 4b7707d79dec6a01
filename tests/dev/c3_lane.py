"""dev aid: C3 (4 096 x 32x32, k = 200) on the lane-per-child kernel under its launch knobs against the 64-row kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
dev = torch.device("cuda", 0)
def engine(**env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env: del os.environ[k_]
    return e
costs, N, M, k = wl.dense_config("c3")
B = costs.shape[0]
d_cost = torch.from_numpy(costs).to(dev)
d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
st = torch.cuda.Stream()
cfgs = [dict()] + [dict(KBEST_FORCE_LANE=1, KBEST_LANE_SPEC=s, KBEST_LANE_NW=w) for w in (2, 4) for s in (8, 12, 16)]
ref = None
for c in cfgs:
    try:
        e = engine(**c)
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        g = d_g.cpu().numpy().view(np.int64)
        if ref is None: ref = g.copy()
        print(f"{str(c):80s} {min(ts[1:]):.3f} ms  same gains {bool((g == ref).all())}", flush=True)
    except Exception as ex:
        print(c, "failed:", ex, flush=True)
