"""Kernel time of B dense 64x64 (k = 200) matrices for the launch-shape knobs in the environment.  Development aid."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
B = int(sys.argv[1]); N = M = int(sys.argv[2]) if len(sys.argv) > 2 else 64; k = 200
eng = pk.KBestEngine(0)
costs = torch.from_numpy(wl.dense_batch(B, N, M, 0x5EED0000 + 1000 * N + k)).to(dev)
r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
e1.record(); torch.cuda.synchronize()
print(f"B={B} {N}x{M} NW={os.environ.get('KBEST_NWAVES','-')} SPEC={os.environ.get('KBEST_SPEC','-')}: {e0.elapsed_time(e1)/5:.3f} ms  gsum {float(g.sum()):.6e}")
