"""C4 on the device entry with the result tables in device memory vs in pinned host memory (zero-copy stores over PCIe), and
the upload alone.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
B, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else B
costs_h = torch.from_numpy(wl.dense_batch(B, N, M, seed)).pin_memory()
costs = costs_h.to(dev)
eng = pk.KBestEngine(0); eng.reserve(B, N, k)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
def bufs(host):
    kw = dict(pin_memory=True) if host else dict(device=dev)
    return (torch.empty((B, k, M), dtype=torch.int32, **kw), torch.empty((B, k, N), dtype=torch.int32, **kw),
            torch.empty((B, k), dtype=torch.float64, **kw), torch.empty(B, dtype=torch.int32, **kw))
for name, host, c4 in (("device tables", False, True), ("host tables (r4c+c4r)", True, True), ("host tables (r4c only)", True, False)):
    r4c, c4r, g, nf = bufs(host)
    for _ in range(2): eng.kbest_dev(costs, B, N, M, k, r4c, c4r if c4 else None, g, nf, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.kbest_dev(costs, B, N, M, k, r4c, c4r if c4 else None, g, nf, stream=s)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/5:.3f} ms  gsum {float(g.sum()):.6e}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): costs.copy_(costs_h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
print(f"H2D of the cost blocks ({costs_h.numel()*8/1e6:.1f} MB, pinned): {e0.elapsed_time(e1)/5:.3f} ms")
