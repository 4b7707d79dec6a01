"""Kernel time of one dense config on the lane-per-child kernel for the knobs in the environment (KBEST_LIB selects a build).
usage: python tests/dev/lane_time.py c2|c3 [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
cfg = sys.argv[1]
Bc, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else Bc
costs = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
eng = pk.KBestEngine(0); eng.reserve(B, N, k)
eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
env = {k_: v for k_, v in os.environ.items() if k_.startswith("KBEST_")}
print(f"{cfg} B={B} {env}: {best:.3f} ms  nf {int(nf.sum())} gsum {g.sum().item():.9e}", flush=True)
