"""How much would a perfect threshold from the first round on save?  The same matrix B times; second run with the
cutoff variant and cutoff = (k-th best - best) * (1 + 1e-9): an upper bound on what any in-round threshold tightening
could gain.  Development aid."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

eng = pk.KBestEngine(0)
dev = torch.device("cuda", 0)
B, N, M, k = 768, 64, 64, 200
base = wl.dense_batch(8, N, M, wl.DENSE_CONFIGS["c4"][4])
for mi in range(4):
    costs = np.tile(base[mi], (B, 1))
    d_cost = torch.from_numpy(costs).to(dev)
    d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_gain = torch.empty((B, k), dtype=torch.float64, device=dev); d_nf = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    def run(cutoff):
        ts = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                e0.record()
                eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, stream=s.cuda_stream, cutoff=cutoff)
                e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts)
    t_plain = run(None)
    g = d_gain[0].cpu().numpy(); nf0 = int(d_nf[0])
    cut = (g[nf0 - 1] - g[0]) * (1 + 1e-9)
    t_cut = run(cut)
    nf1 = int(d_nf[0])
    print(f"matrix {mi}: plain {t_plain:.3f} ms (nf {nf0}), perfect threshold from the start {t_cut:.3f} ms (nf {nf1}): {100*(1-t_cut/t_plain):.1f}% less; gap {cut:.4f}")
    for f in (1.5, 2, 3, 5, 10, 20):  # a looser a-priori bound (e.g. the k-th cheapest 2-swap of the optimum: 10-20x the gap)
        t = run(cut * f)
        print(f"    threshold {f:4.1f} x gap: {t:.3f} ms ({100*(1-t/t_plain):.1f}% less)")
