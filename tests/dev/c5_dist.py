"""Diagnostic (GPU box, PROFILE build): distribution over the C5 frames of the small-problem kernel's per-frame lifetime.
KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=4 python3 tests/dev/c5_dist.py [F]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("KBEST_LIB", "libkbest_amd_prof.so")
import numpy as np
import torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nw = int(os.environ.get("KBEST_SMALL_NW", "4"))
k = 200
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
frames = wl.kitti_like_frames(B)
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
prof = torch.zeros((B, 16), dtype=torch.int64, device=dev)
eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
s = torch.cuda.Stream()
for it in range(3):
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
life = p[:, 15] / nw
rounds = p[:, 2] / nw
passes = p[:, 11]
print(f"F={B} NW={nw} kernel {ms:.3f} ms; rows kept: mean {nrow.mean():.1f} min {nrow.min()} max {nrow.max()}")
q = np.percentile(life, [0, 10, 50, 90, 99, 100])
print("per-frame lifetime (cycles per wave): min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f mean %.0f" % (*q, life.mean()))
print("  max lifetime / kernel time = %.0f cycles / %.3f ms = %.2f GHz if the longest frame spanned the launch" % (q[-1], ms, q[-1] / ms / 1e6))
print("rounds: mean %.1f p90 %.1f max %.0f; passes mean %.1f max %.0f" % (rounds.mean(), np.percentile(rounds, 90), rounds.max(), passes.mean(), passes.max()))
for name, x in (("rows kept", nrow), ("rounds", rounds), ("child passes", passes), ("nf", d_nf.cpu().numpy())):
    print(f"  corr(lifetime, {name}) = {np.corrcoef(life, x.astype(np.float64))[0, 1]:.3f}")
order = np.argsort(-life)[:8]
print("slowest frames:", [(int(i), int(life[i]), int(nrow[i]), int(rounds[i]), int(passes[i])) for i in order])
