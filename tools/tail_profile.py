"""Diagnostic: the timeline of ONE launch of the 64-row kernel (needs `make -C probabilisticsemslam_amd/csrc PROFILE=1`): start / end
of every workgroup on the 100 MHz wall clock and the CU it ran on.  Prints the lifetime histogram, the idle share of the CU slots
(makespan x slots - sum of lifetimes) and what list scheduling would do with the same lifetimes in longest-first order.
Run on the GPU box:  python tools/tail_profile.py [config] [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KBEST_LIB", "libkbest_amd_prof.so")
import heapq
import numpy as np
import torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
Bc, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else Bc
d_cost = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
d_nf = torch.empty(B, dtype=torch.int32, device=dev)
prof = torch.zeros(B * 21, dtype=torch.int64, device=dev)
eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
s = torch.cuda.Stream()
for it in range(3):
    prof.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record()
        eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, stream=s.cuda_stream, tie_check=False)
        e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
p = prof.cpu().numpy()
t = p[B * 16:].reshape(B, 5)
start, end = (t[:, 0] - t[:, 0].min()) / 100.0, (t[:, 1] - t[:, 0].min()) / 100.0  # microseconds
life = end - start
hw = t[:, 2]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)  # (XCC, SE, CU): a label per CU
makespan = end.max()
ncu = len(np.unique(cu))
slots = int(round((life.sum() / makespan) + 0.5))
print(f"{cfg}: {B} matrices, launch {ms:.3f} ms (events), makespan {makespan:.1f} us on the wall clock, {ncu} CUs seen")
q = np.percentile(life, [0, 5, 25, 50, 75, 95, 100])
print("lifetime of a workgroup (us): min %.0f  p5 %.0f  p25 %.0f  median %.0f  p75 %.0f  p95 %.0f  max %.0f   mean %.0f" % (*q, life.mean()))
first = start < 5.0
print(f"first generation (started within 5 us): {first.sum()} workgroups, mean lifetime {life[first].mean():.0f} us; later ones: {(~first).sum()}, mean {life[~first].mean():.0f} us")
nslot = int(first.sum())
busy = life.sum()
print(f"slots = {nslot}: slot-time {nslot * makespan:.0f} us, busy {busy:.0f} us -> idle {100 * (1 - busy / (nslot * makespan)):.1f} %;  sum of lifetimes / slots = {busy / nslot:.1f} us (the makespan of a perfect packing)")
last = np.sort(end)[::-1]
print("the last workgroups end at (us):", np.round(last[:8], 1), "; the slot that ends first after which nothing starts:", round(float(np.sort(end)[-nslot]), 1))
def list_schedule(order):
    h = [0.0] * nslot
    heapq.heapify(h)
    for i in order:
        t0 = heapq.heappop(h)
        heapq.heappush(h, t0 + life[i])
    return max(h)
print(f"list scheduling of these lifetimes on {nslot} slots: launch order {list_schedule(range(B)):.1f} us, longest first {list_schedule(np.argsort(-life)):.1f} us, shortest first {list_schedule(np.argsort(life)):.1f} us")
hist, edges = np.histogram(life, bins=12)
for h, a, b in zip(hist, edges[:-1], edges[1:]):
    print(f"  {a:7.0f} - {b:7.0f} us  {h:5d} {'#' * int(60 * h / hist.max())}")
# correlation of a matrix' lifetime with the number of solutions beyond the optimum it had to look at is what a predictor would need
print("per CU: workgroups", np.bincount(np.unique(cu, return_inverse=True)[1]).tolist()[:16], "...")
if os.environ.get("TAIL_FEATURES"):
    # what could PREDICT a matrix' lifetime before it runs?  Features of the root solution (duals from the assign entry).
    costs = wl.dense_batch(B, N, M, seed)
    os.environ["KBEST_LIB"] = "libkbest_amd.so"
    from probabilisticsemslam_amd import engine as EM
    EM._lib = None
    e2 = pk.KBestEngine(0)
    ok, r4c, c4r, g, u, v = e2.assign(costs, N, M)
    Cm = costs.reshape(B, M, N)                      # [b][col][row]
    R = Cm - u[:, :, None] - v[:, None, :]            # reduced costs >= 0, 0 on the optimum
    Rs = np.sort(R, axis=2)                           # per column ascending
    second = Rs[:, :, 1]                              # cheapest alternative of every column
    feats = {
        "sum of the columns' second smallest reduced cost": second.sum(axis=1),
        "8th smallest second-smallest": np.sort(second, axis=1)[:, 7],
        "16th smallest": np.sort(second, axis=1)[:, 15],
        "entries below 0.02": (R < 0.02).sum(axis=(1, 2)),
        "entries below 0.05": (R < 0.05).sum(axis=(1, 2)),
        "entries below 0.10": (R < 0.10).sum(axis=(1, 2)),
        "the k-th gain's gap (the answer)": d_gain.cpu().numpy()[:, k - 1] - d_gain.cpu().numpy()[:, 0],
        "children started [11]": p[:B * 16].reshape(B, 16)[:, 5].astype(float),
        "whole kernel wave-cycles [13]": p[:B * 16].reshape(B, 16)[:, 13].astype(float),
    }
    for name, f in feats.items():
        c = np.corrcoef(f, life)[0, 1]
        pred_order = np.argsort(-f if c > 0 else f)
        print(f"  corr(lifetime, {name}) = {c:+.3f}; list scheduling in that order: {list_schedule(pred_order):.1f} us")
    np.savez(os.path.join(ROOT, "gpurun_out", f"tail_{cfg}.npz"), life=life, start=start, end=end, cu=cu, prof=p[:B * 16].reshape(B, 16))
