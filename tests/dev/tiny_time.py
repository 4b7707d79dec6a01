"""Diagnostic (GPU box): one-frame-per-call latency of the association entry on small frames, tiny kernel against the enumeration
kernel (KBEST_NO_TINY), KITTI-like and DENSE (every entry within the gate: nothing pruned by +inf) cost blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

def timeit(eng, frames, nL, nM, k=200, n=200):
    for i in range(-50, n):
        if i == 0:
            t0 = time.perf_counter()
        eng.weights([frames[i % len(frames)]], [nL], [nM], k, condition=True)
    return 1e6 * (time.perf_counter() - t0) / n

tiny = pk.KBestEngine(0)
os.environ["KBEST_NO_TINY"] = "1"
plain = pk.KBestEngine(0)
rng = np.random.default_rng(5)
for nL, nM in ((6, 3), (6, 5), (12, 5), (20, 4), (10, 6), (40, 3), (6, 6), (20, 5), (50, 4), (30, 4), (14, 6)):
    fr = wl.kitti_like_frames(32, nL=nL, nM=nM, seed=77 + nL)
    nR = nL + nM
    dense = []
    for _ in range(8):
        C = np.full(nR * nM, np.inf)
        for c in range(nM):
            C[c * nR: c * nR + nL] = rng.random(nL) * 30.0
            C[c * nR + nL + c] = 10.0
        dense.append(C)
    cnt = 1
    for c in range(nM):
        cnt *= nR - c
    print(f"nL={nL:2d} nM={nM} assignments {cnt:8d}: KITTI-like tiny {timeit(tiny, fr, nL, nM):7.1f} us, enumeration {timeit(plain, fr, nL, nM):7.1f} us | "
          f"dense tiny {timeit(tiny, dense, nL, nM):7.1f} us, enumeration {timeit(plain, dense, nL, nM):7.1f} us (python call overhead included)")
