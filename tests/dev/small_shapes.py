"""Diagnostic (GPU box): the small-problem kernel's waves-per-problem rule on rectangular batches through kbest_batch_f64_dev:
kernel ms for KBEST_SMALL_NW = rule / 4 / 8 (run once per setting).  python3 tests/dev/small_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import probabilisticsemslam_amd as pk

dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
rng = np.random.default_rng(3)
out = []
for (B, N, M, k) in ((1024, 28, 10, 200), (2048, 20, 8, 50), (4096, 12, 6, 100), (1024, 32, 16, 200), (600, 32, 24, 200), (3000, 16, 4, 20), (800, 24, 12, 400)):
    costs = rng.random((B, M, N)) * 30.0
    d_cost = torch.from_numpy(costs).to(dev)
    d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
    d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_nf = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    ts = []
    for it in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
            eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, stream=s.cuda_stream)
            e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    out.append(f"{B}x{N}x{M},k={k}: {np.mean(ts):.3f}")
print("NW=%s  " % os.environ.get("KBEST_SMALL_NW", "rule") + "  ".join(out))
