"""Does the ORDER of the columns matter to the enumeration's cost?  Murty's partition fixes the columns before c in the child on
column c: with the columns that are dear to change first and the cheap ones last, the children that carry the k best have nearly
everything fixed.  Probe: permute the columns of every matrix on the host by the root's first-step lower bound (descending /
ascending / none) and time the unchanged engine.  usage: python tests/dev/colorder_probe.py c4|c3|c2"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
cfg = sys.argv[1]
B, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
costs = wl.dense_batch(B, N, M, seed)            # (B, N*M) column-major blocks: C[r + c*N]
eng = pk.KBestEngine(0)
ok, r4c, c4r, g, u, v = eng.assign(costs, N, M, shift=False)
C = costs.reshape(B, M, N)                        # [b, c, r]
rc = C - u[:, :, None] - v[:, None, :]            # reduced costs [b, c, r]
own = np.zeros_like(rc, dtype=bool)
bi, ci = np.meshgrid(np.arange(B), np.arange(M), indexing="ij")
own[bi, ci, r4c] = True
m = np.where(own, np.inf, rc).min(axis=2)         # [b, c]: cheapest other row of column c
def run(perm_kind):
    if perm_kind == "none": P = np.tile(np.arange(M), (B, 1))
    elif perm_kind == "dear_first": P = np.argsort(-m, axis=1, kind="stable")
    elif perm_kind == "cheap_first": P = np.argsort(m, axis=1, kind="stable")
    Cp = np.take_along_axis(C, P[:, :, None], axis=1)
    d_cost = torch.from_numpy(np.ascontiguousarray(Cp.reshape(B, N * M))).to(dev)
    o = (torch.empty((B, k, M), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    eng.reserve(B, N, k)
    eng.kbest_dev(d_cost, B, N, M, k, *o); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.kbest_dev(d_cost, B, N, M, k, *o)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best, o[2].sum().item()
for kind in ("none", "dear_first", "cheap_first", "none"):
    t, gs = run(kind)
    print(f"{cfg} columns {kind:12s}: {t:.3f} ms   (sum of gains {gs:.6f})", flush=True)
