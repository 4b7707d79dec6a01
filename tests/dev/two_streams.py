"""Steady-state throughput with ONE batch in flight (one context, one stream) against TWO (two contexts, two streams,
batches alternating): the second batch's workgroups fill the CUs that the first one's last generation leaves idle.
usage: python tests/dev/two_streams.py c4|c3|c2 [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
cfg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
costs = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
def bufs():
    return (torch.empty((B, k, M), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
            torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
engs = [pk.KBestEngine(0) for _ in range(3)]
strs = [torch.cuda.Stream(device=dev) for _ in range(3)]
outs = [bufs() for _ in range(3)]
for e in engs: e.reserve(B, N, k)
def run(n):
    for i in range(n): engs[i].kbest_dev(costs, B, N, M, k, *outs[i], stream=strs[i].cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps): engs[i % n].kbest_dev(costs, B, N, M, k, *outs[i % n], stream=strs[i % n].cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return 1e3 * best
for n in (1, 2, 3, 1, 2):
    print(f"{cfg}: {n} in flight: {run(n):.3f} ms per batch", flush=True)
same = all(torch.equal(outs[0][j], outs[1][j]) for j in range(4))
print("same results", same)
