"""dev aid: what enumerating k + 1 instead of k costs (the boundary-tie check), interleaved on one box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
dev = torch.device("cuda", 0)
e = pk.KBestEngine(0)
st = torch.cuda.Stream()
for cfg in ("c4", "c3", "c2"):
    costs, N, M, k0 = wl.dense_config(cfg)
    B = costs.shape[0]
    d_cost = torch.from_numpy(costs).to(dev)
    res = {}
    bufs = {}
    for k in (k0, k0 + 1):
        bufs[k] = (torch.empty((B, k, N), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
                   torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
        e.reserve(B, N, k)
    for rnd in range(3):
        for k in (k0, k0 + 1):
            d_r, d_c, d_g, d_n = bufs[k]
            ts = []
            for it in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res.setdefault(k, []).append(float(np.mean(ts[1:])))
    print(cfg, {k: [round(x, 4) for x in v] for k, v in res.items()}, flush=True)
