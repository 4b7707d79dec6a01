"""Diagnostic (PROFILE build): the timeline of a RELAY launch of the 64-row kernel -- start / end of every (piece, matrix)
workgroup: lifetimes per piece, the gap between a piece's end and the next piece's start, idle share of the slots.
Run on the GPU box:  [KBEST_RELAY=P [KBEST_RELAY_FIRST=.. KBEST_RELAY_STEP=..]] python tools/relay_profile.py [config]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KBEST_LIB", "libkbest_amd_prof.so")
import numpy as np
import torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
if "KBEST_RELAY" not in os.environ:  # the launch plan's own choice for the bench configs (kbest_capi.cpp, relay_plan): the record layout depends on it
    os.environ["KBEST_RELAY"] = "3"
    os.environ.setdefault("KBEST_RELAY_FIRST", "384" if cfg == "c4" else "640")
    os.environ.setdefault("KBEST_RELAY_STEP", "384" if cfg == "c4" else "256")
P = int(os.environ["KBEST_RELAY"])
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
Bc, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
B = Bc
d_cost = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
d_nf = torch.empty(B, dtype=torch.int32, device=dev)
W = B * max(P, 1)
prof = torch.zeros(W * 21, dtype=torch.int64, device=dev)
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
t = prof.cpu().numpy()[W * 16:].reshape(max(P, 1), B, 5)
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
start, end = (t[:, :, 0] - t0) / 100.0, (t[:, :, 1] - t0) / 100.0
print(f"{cfg}, relay {P}: launch {ms:.3f} ms, makespan {end.max():.0f} us")
for j in range(max(P, 1)):
    life = end[j] - start[j]
    line = f"  piece {j}: start mean {start[j].mean():7.0f} (min {start[j].min():6.0f}, max {start[j].max():6.0f})  lifetime mean {life.mean():6.1f} p95 {np.percentile(life, 95):6.1f} max {life.max():6.1f}"
    if j > 0:
        gap = start[j] - end[j - 1]
        line += f"   gap to the piece before: mean {gap.mean():7.1f} min {gap.min():7.1f} (negative: it waited {(-gap[gap < 0]).sum():.0f} us in all, {int((gap < 0).sum())} workgroups)"
    print(line)
if P > 1:
    tin, tout = (t[:, :, 3] - t0) / 100.0, (t[:, :, 4] - t0) / 100.0
    for j in range(P):
        m_in = t[j, :, 3] > 0
        m_out = t[j, :, 4] > 0
        msg = f"  piece {j}:"
        if m_in.any():
            msg += f" start -> image in LDS: mean {(tin[j] - start[j])[m_in].mean():6.2f} us (max {(tin[j] - start[j])[m_in].max():6.1f});"
        if m_out.any():
            msg += f" rounds over -> end (image out, L2 written back, flag): mean {(end[j] - tout[j])[m_out].mean():6.2f} us (max {(end[j] - tout[j])[m_out].max():6.1f}), {int(m_out.sum())} hand-overs"
        print(msg)
acc = prof.cpu().numpy()[:B * 16].reshape(B, 16).astype(np.float64)
names = ["setup+root", "B busy", "merge", "A/D busy", "4", "5", "6", "rounds", "8", "9", "finish", "wait after B", "wait after A", "whole (lane-0 sum over waves)", "round prologue", "filter"]
print("  accumulators, mean per matrix (cycle-counter ticks, summed over the waves' lane 0 and over the pieces): " + ", ".join(f"{n} {acc[:, i].mean():.0f}" for i, n in enumerate(names)))
grid = np.arange(0.0, end.max() + 50.0, 50.0)
st_, en_ = np.sort(start.ravel()), np.sort(end[end > 0].ravel())
active = [int(np.searchsorted(st_, x, side="right") - np.searchsorted(en_, x, side="right")) for x in grid]
print("  resident workgroups every 50 us: " + " ".join(str(a) for a in active))
hw = t[:, :, 2].ravel()
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)  # (XCC, SE, CU)
gaps = []
for c in np.unique(cu):
    m = cu == c
    en = np.sort(end.ravel()[m]); stl = np.sort(start.ravel()[m]); stl = stl[stl > 5.0]
    n = min(len(en), len(stl))
    gaps.extend((stl[:n] - en[:n]).tolist())   # the j-th later start of a CU follows its j-th end
gaps = np.array(gaps)
print(f"  a slot between two workgroups (end stamp -> next start stamp on that CU, {len(gaps)} turnovers): median {np.median(gaps):.1f} us, mean {gaps.mean():.1f}, p90 {np.percentile(gaps, 90):.1f}, p99 {np.percentile(gaps, 99):.1f}")
busy = (end - start).sum()
nslot = int((start[0] < 5.0).sum())
print(f"  slots {nslot}: busy {busy:.0f} us of {nslot * end.max():.0f} -> idle {100 * (1 - busy / (nslot * end.max())):.1f} %; sum of all lifetimes / slots = {busy / nslot:.0f} us")
