"""Diagnostic (GPU box): kernel ms of F conditioned KITTI-like frames (kBest2DCutoff(200, 42) on the small-problem kernel) over F,
for the launch-shape rule of kbest_capi.cpp (small_waves).  KBEST_SMALL_NW=<n> forces a shape.
python3 tests/dev/c5_sweep.py 300 600 768 900 1000 1024 1200 2000"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

Fs = [int(a) for a in sys.argv[1:]] or [1000]
k = 200
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
allf = wl.kitti_like_frames(max(Fs))
out = []
for B in Fs:
    frames = allf[:B]
    conds, idxs = eng.condition_costs(frames, [30] * B, [10] * B)
    nrow = np.array([len(i) for i in idxs], np.int32)
    N, M = int(nrow.max()), 10
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(nrow[:-1].astype(np.int64) * M)
    d_cost = torch.from_numpy(np.concatenate(conds)).to(dev)
    kw = dict(cutoff=42.0, d_nRow=torch.from_numpy(nrow).to(dev), d_nCol=torch.full((B,), M, dtype=torch.int32, device=dev),
              d_costOff=torch.from_numpy(off).to(dev))
    d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
    d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_nf = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
            eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, stream=s.cuda_stream, **kw)
            e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1))
    out.append("%d: %.3f" % (B, float(np.mean(ts))))
print("NW=%s  F: ms  " % os.environ.get("KBEST_SMALL_NW", "rule") + "  ".join(out))
