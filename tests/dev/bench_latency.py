"""Per-call latency of the host-pointer entry points on small batches (the reference calls assignmentProb once per
frame).  Development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol

eng = pk.KBestEngine(0)
frames = wl.kitti_like_frames(64, nL=20, nM=10)
conds = []
for f in frames:
    c, idx = ol.condition_costs(f, 30, 10)
    conds.append((c, len(idx) - 10))
for B in (1, 4, 16, 64):
    cs = [c for c, _ in conds[:B]]; nL = [l for _, l in conds[:B]]
    eng.weights(cs, nL, [10] * B, 200)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        eng.weights(cs, nL, [10] * B, 200)
    dt = (time.perf_counter() - t0) / n
    print(f"assignmentProb batch B={B}: {dt*1e3:.3f} ms per call, {dt*1e6/B:.1f} us per frame", flush=True)
t0 = time.perf_counter()
for c, l in conds[:32]:
    ol.assignment_prob(c, l, 10, 200)
print(f"CPU checker (1 core): {(time.perf_counter()-t0)/32*1e6:.1f} us per frame")
costs = np.random.default_rng(1).random((1, 64 * 64))
eng.kbest(costs, 64, 64, 200)
t0 = time.perf_counter()
for _ in range(20):
    eng.kbest(costs, 64, 64, 200)
print(f"kBest2D 64x64 k=200, B=1: {(time.perf_counter()-t0)/20*1e3:.3f} ms per call")
# the solver alone on the same frames (what the weights epilogue adds)
cs = [c for c, _ in conds[:1]]; nL1 = conds[0][1]
c0 = np.asarray(cs[0]).reshape(1, -1)
eng.kbest(c0, nL1 + 10, 10, 200, cutoff=42.0)
t0 = time.perf_counter()
for _ in range(20):
    eng.kbest(c0, nL1 + 10, 10, 200, cutoff=42.0)
print(f"kBest2DCutoff on the same frame, B=1: {(time.perf_counter()-t0)/20*1e3:.3f} ms per call")
