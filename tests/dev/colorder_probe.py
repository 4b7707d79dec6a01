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
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); S = ts.cuda_stream
ok, r4c, c4r, g, u, v = eng.assign(costs, N, M, shift=False)
C = costs.reshape(B, M, N)                        # [b, c, r]
rc = C - u[:, :, None] - v[:, None, :]            # reduced costs [b, c, r]
own = np.zeros_like(rc, dtype=bool)
bi, ci = np.meshgrid(np.arange(B), np.arange(M), indexing="ij")
own[bi, ci, r4c] = True
m = np.where(own, np.inf, rc).min(axis=2)         # [b, c]: cheapest other row of column c
fr = r4c                                           # [b, c] row of column c
rin = np.take_along_axis(rc, fr[:, None, :].repeat(M, axis=1), axis=2) if False else None
# last arc into the freed row fr = r4c[c] from any other column j: rc[b, j, fr[b, c]]
rcT = np.transpose(rc, (0, 2, 1))                 # [b, r, j]
into = np.take_along_axis(rcT, fr[:, :, None].repeat(M, axis=2), axis=1)  # [b, c, j] = rc[b, j, fr[b,c]]
into[bi, ci, ci] = np.inf
min_in = np.clip(into.min(axis=2), 0, None)       # [b, c]
exact = None
if os.environ.get("PROBE_EXACT"):
    from scipy.optimize import linear_sum_assignment
    nb = min(B, int(os.environ.get("PROBE_EXACT")))
    exact = m + min_in
    exact = exact.copy()
    for b in range(nb):
        Cb = C[b].T.copy()   # [r, c]
        base = Cb[r4c[b], np.arange(M)].sum()
        for c in range(M):
            r = r4c[b, c]; old = Cb[r, c]; Cb[r, c] = 1e9
            ri, cj = linear_sum_assignment(Cb)
            exact[b, c] = Cb[ri, cj].sum() - base
            Cb[r, c] = old
freq = None
if os.environ.get("PROBE_FREQ"):
    # the ideal static order, from the answer itself: how often each column differs from the optimum in the k best
    nf0, r0, c0, g0 = eng.kbest(costs, N, M, k)
    freq = (r0 != r0[:, :1, :]).sum(axis=1).astype(np.float64)   # [b, c]
def run(perm_kind):
    if perm_kind == "none": P = np.tile(np.arange(M), (B, 1))
    elif perm_kind == "dear_first": P = np.argsort(-m, axis=1, kind="stable")
    elif perm_kind == "cheap_first": P = np.argsort(m, axis=1, kind="stable")
    elif perm_kind == "exact_first": P = np.argsort(-exact, axis=1, kind="stable")
    elif perm_kind == "freq_last": P = np.argsort(freq, axis=1, kind="stable")            # rarely changed first, often changed last
    elif perm_kind == "freq_then_exact":                                                   # never-changed columns by dearness, the rest by frequency
        keyf = np.where(freq == 0, -1e9 - (exact if exact is not None else m), freq)
        P = np.argsort(keyf, axis=1, kind="stable")
    elif perm_kind.startswith("clip"):   # exact keys only up to the sum of the m cheapest: beyond it, order by the two-arc bound
        mm = int(perm_kind[4:])
        bnd = np.sort(exact, axis=1)[:, :mm].sum(axis=1, keepdims=True)
        keyc = np.where(exact <= bnd, exact, bnd + (m + min_in))
        P = np.argsort(-keyc, axis=1, kind="stable")
    elif perm_kind == "dear2_first": P = np.argsort(-(m + min_in), axis=1, kind="stable")
    elif perm_kind == "dearmax_first": P = np.argsort(-np.maximum(m, min_in), axis=1, kind="stable")
    Cp = np.take_along_axis(C, P[:, :, None], axis=1)
    d_cost = torch.from_numpy(np.ascontiguousarray(Cp.reshape(B, N * M))).to(dev)
    o = (torch.empty((B, k, M), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
    eng.reserve(B, N, k)
    eng.kbest_dev(d_cost, B, N, M, k, *o, stream=S); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.kbest_dev(d_cost, B, N, M, k, *o, stream=S)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best, o[2].sum().item()
for kind in (("exact_first", "freq_last", "freq_then_exact", "exact_first") if (exact is not None and freq is not None) else ("exact_first", "clip4", "clip6", "clip8", "clip12", "exact_first") if exact is not None else ("none", "dear_first", "dear2_first", "dearmax_first", "none")):
    t, gs = run(kind)
    print(f"{cfg} columns {kind:12s}: {t:.3f} ms   (sum of gains {gs:.6f})", flush=True)
