"""Diagnostic: per-phase cycle shares of the k-best kernel (needs `make -C probabilisticsemslam_amd/csrc PROFILE=1`).
Run on the GPU box:  KBEST_LIB=libkbest_amd_prof.so python tools/phase_profile.py [config] [B]"""
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
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
kw = {}
if cfg == "c5":  # conditioned KITTI-like frames, kBest2DCutoff(42) as assignmentProb calls it
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    k = 200
    frames = wl.kitti_like_frames(B)
    conds, idxs = eng.condition_costs(frames, [30] * B, [10] * B)
    nrow = np.array([len(i) for i in idxs], np.int32)
    N, M = int(nrow.max()), 10
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(nrow[:-1].astype(np.int64) * M)
    d_cost = torch.from_numpy(np.concatenate(conds)).to(dev)
    kw = dict(cutoff=42.0, d_nRow=torch.from_numpy(nrow).to(dev), d_nCol=torch.full((B,), M, dtype=torch.int32, device=dev),
              d_costOff=torch.from_numpy(off).to(dev))
else:
    Bc, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else Bc
    costs = wl.dense_batch(B, N, M, seed)
    d_cost = torch.from_numpy(costs).to(dev)
nw = int(os.environ.get("KBEST_NWAVES", "8"))
d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
d_nf = torch.empty(B, dtype=torch.int32, device=dev)
prof = torch.zeros((B, 16), dtype=torch.int64, device=dev)
eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
s = torch.cuda.Stream()
for it in range(2):
    prof.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record()
        eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, stream=s.cuda_stream, **kw)
        e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
p = prof.cpu().numpy().astype(np.float64)
m = p.mean(axis=0)
# (the library's routing rule, kbest_capi.cpp: small-problem kernel for rectangular or chip-underfilling batches of <= 32 rows)
small = N <= 32 and not os.environ.get("KBEST_NO_SMALL") and (os.environ.get("KBEST_FORCE_SMALL") or M < N or (B <= 512 and kw))
if small:
    nw = int(os.environ.get("KBEST_SMALL_NW", "16" if B <= 256 else ("8" if B <= 1536 else "4")))
    names = ["tile set-up", "root (+barrier)", "rounds", "select+emission", "node load", "filter", "child dijkstra", "finish completed",
             "wait after children", "wait after filter", "rows scanned", "child passes", "merge", "wait after merge", "epilogue", "kernel cyc (sum over waves)"]
    tot = m[15]
    print(f"{cfg} B={B} small kernel NW={nw} kernel {ms:.3f} ms (profiled build)")
    for i, n in enumerate(names):
        extra = f"  = {100*m[i]/tot:5.1f}% of wave-cycles, {m[i]/nw:9.0f} cyc/wave" if i not in (2, 10, 11) else ""
        print(f"  [{i:2d}] {n:28s} {m[i]:14.1f}{extra}")
    rounds = m[2] / nw
    print(f"  rounds {rounds:.1f}; per wave kernel cycles {tot/nw:.0f} = {tot/nw/max(rounds,1):.0f} per round; dijkstra cycles per pass {m[6]/max(m[11],1):.0f}; rows scanned per pass {m[10]/max(m[11],1):.1f}")
    print("  nf:", d_nf.cpu().numpy()[:8])
    sys.exit(0)
if os.environ.get("KBEST_FORCE_LANE"):  # lane-per-child kernel (kbest_lane.hip)
    nw = int(os.environ.get("KBEST_LANE_NW", "1" if N <= 16 else "2"))
    names = ["setup+root", "stepping busy", "merge(+barriers)", "select/emit/node loads", "children started", "lane-steps (active children x passes)",
             "children completed", "rounds (x waves)", "-", "passes", "cyc finishing completed", "wait after stepping", "wait after A", "kernel cyc (sum over waves)",
             "round prologue", "wait after finishing"]
    tot = m[13]
    print(f"{cfg} B={B} lane kernel NW={nw} SPEC={os.environ.get('KBEST_LANE_SPEC','-')} kernel {ms:.3f} ms (profiled build)")
    for i, n in enumerate(names):
        extra = f"  = {100*m[i]/tot:5.1f}% of wave-cycles, {m[i]/nw:9.0f} cyc/wave" if i in (0, 1, 2, 3, 10, 11, 12, 14, 15) else ""
        print(f"  [{i:2d}] {n:40s} {m[i]:14.1f}{extra}")
    rounds = m[7] / nw
    stepc = m[1]
    print(f"  rounds {rounds:.1f}; kernel cycles per wave {tot/nw:.0f} = {tot/nw/max(rounds,1):.0f} per round; passes per round per wave {m[9]/nw/max(rounds,1):.1f}; "
          f"cycles per pass {stepc/max(m[9],1):.0f}; lanes active per pass {m[5]/max(m[9],1):.1f}; steps per child {m[5]/max(m[4],1):.2f}; completed/started {m[6]/max(m[4],1):.2f}")
    print("  nf:", d_nf.cpu().numpy()[:8])
    sys.exit(0)
names = ["setup+root", "B busy", "C merge(+barrier)", "A/D busy", "children started", "child steps", "children completed",
         "rounds", "cyc in child dijkstra", "cyc child set-up", "cyc flip+gain", "wait after B", "wait after A", "kernel cyc (sum over waves)",
         "round prologue", "first-step filter busy"]
print(f"{cfg} B={B} NW={nw} SPEC={os.environ.get('KBEST_SPEC','4')} kernel {ms:.3f} ms (profiled build)")
tot = m[13]
for i, n in enumerate(names):
    extra = ""
    if i in (0, 1, 2, 3, 8, 9, 10, 11, 12, 14, 15):
        extra = f"  = {100*m[i]/tot:5.1f}% of wave-cycles"
    print(f"  [{i:2d}] {n:28s} {m[i]:14.1f}{extra}")
print(f"  per wave kernel cycles {tot/nw:.0f}; cycles/step in child dijkstra {m[8]/max(m[5],1):.0f}; set-up cycles/child {m[9]/max(m[4],1):.0f}; "
      f"flip+gain cycles/completed child {m[10]/max(m[6],1):.0f}; steps/child {m[5]/max(m[4],1):.2f}")
print("  nf:", d_nf.cpu().numpy()[:8])
w = p[:, 13] / nw  # per-matrix lifetime (cycles, mean over its waves)
print(f"  per-matrix lifetime: mean {w.mean():.0f} std {w.std():.0f} ({100*w.std()/w.mean():.1f}%) min {w.min():.0f} max {w.max():.0f} (max/mean {w.max()/w.mean():.2f}); "
      f"steps: mean {p[:,5].mean():.0f} std {p[:,5].std():.0f} max {p[:,5].max():.0f}; rounds mean {p[:,7].mean()/nw:.1f} max {p[:,7].max()/nw:.0f}")
