"""Per-call latency distribution of kbest_assoc_probs_batch_f64(B=1) over the C5 frames (dev aid)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
F = 400
frames = wl.kitti_like_frames(F)
p = lambda a: a.ctypes.data_as(C.c_void_p)
one_l, one_m, zero = np.array([20], np.int32), np.array([10], np.int32), np.zeros(1, np.int64)
op, onf = np.zeros(10 * 21), np.zeros(1, np.int32)
lat = np.zeros((3, F))
for rep in range(3):
    for i in range(F):
        t = time.perf_counter()
        eng.lib.kbest_assoc_probs_batch_f64(eng.ctx, 1, p(one_l), p(one_m), p(frames[i]), p(zero), 200, p(op), p(zero), p(onf))
        lat[rep, i] = time.perf_counter() - t
l = lat[1:].mean(axis=0) * 1e6
print("mean %.1f median %.1f p95 %.1f max %.1f us" % (l.mean(), np.median(l), np.percentile(l, 95), l.max()))
order = np.argsort(-l)[:12]
print("slowest frames:", [(int(i), round(float(l[i]), 1), round(float(lat[1, i] * 1e6), 1), round(float(lat[2, i] * 1e6), 1)) for i in order])
h, e = np.histogram(l, bins=[0, 150, 175, 200, 225, 250, 300, 400, 600, 1000, 1e9])
print("hist:", list(zip(e[:-1].astype(int).tolist(), h.tolist())))
