"""Column-order probe on C5-like rectangular frames (conditioned 28x10 blocks, k = 200, cutoff 42): columns permuted on the host
by the exact cost of changing each column (scipy), engine unchanged.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scipy.optimize import linear_sum_assignment
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
F, k = 3000, 200
eng = pk.KBestEngine(0)
frames = wl.kitti_like_frames(F)
conds, idxs = eng.condition_costs(frames, [30] * F, [10] * F)
nrow = np.array([len(i) for i in idxs])
N = int(np.bincount(nrow).argmax()); M = 10
sel = [b for b in range(F) if nrow[b] == N][:1000]
B = len(sel)
C = np.stack([conds[b].reshape(M, N) for b in sel])     # [b, c, r]
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); S = ts.cuda_stream
key = np.zeros((B, M))
for b in range(B):
    Cb = C[b].T.copy()                                    # [r, c]
    ri, cj = linear_sum_assignment(Cb); base = Cb[ri, cj].sum()
    r4c = np.empty(M, int); r4c[cj] = ri
    for c in range(M):
        old = Cb[r4c[c], c]; Cb[r4c[c], c] = 1e9
        r2, c2 = linear_sum_assignment(Cb); key[b, c] = Cb[r2, c2].sum() - base
        Cb[r4c[c], c] = old
def run(kind):
    P = np.tile(np.arange(M), (B, 1)) if kind == "none" else np.argsort(-key if kind == "dear_first" else key, axis=1, kind="stable")
    Cp = np.take_along_axis(C, P[:, :, None], axis=1)
    d_cost = torch.from_numpy(np.ascontiguousarray(Cp.reshape(B, N * M))).to(dev)
    o = (torch.empty((B, k, M), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    eng.reserve(B, N, k)
    eng.kbest_dev(d_cost, B, N, M, k, *o, cutoff=42.0, stream=S); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.kbest_dev(d_cost, B, N, M, k, *o, cutoff=42.0, stream=S)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best, int(o[3].sum().item())
for kind in ("none", "dear_first", "cheap_first", "none"):
    t, nf = run(kind)
    print(f"{B} frames {N}x{M} columns {kind:12s}: {t:.3f} ms  nf {nf}", flush=True)
