"""Association on mid-size frames (conditioned block of 33..64 rows: general pipeline).  Development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
for nL, nM, F in ((50, 14, 64), (50, 14, 512), (40, 12, 512)):
    fr = wl.kitti_like_frames(F, nL=nL, nM=nM, seed=0xD00D01)
    eng.weights(fr, [nL] * F, [nM] * F, 200, condition=True)
    t0 = time.perf_counter()
    for _ in range(3):
        out, nf = eng.weights(fr, [nL] * F, [nM] * F, 200, condition=True)
    dt = (time.perf_counter() - t0) / 3
    print(f"nL={nL} nM={nM} F={F}: {dt*1e3:.2f} ms per call, {dt*1e6/F:.1f} us per frame, nf mean {nf.mean():.0f}")
