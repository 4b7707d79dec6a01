import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
frames = wl.kitti_like_frames(400)
p = lambda a: a.ctypes.data_as(C.c_void_p)
one_l, one_m, zero = np.array([20], np.int32), np.array([10], np.int32), np.zeros(1, np.int64)
op, onf = np.zeros(10 * 21), np.zeros(1, np.int32)
for rep in range(3):
    slow = []
    for i in range(400):
        t = time.perf_counter()
        rc = eng.lib.kbest_assoc_probs_batch_f64(eng.ctx, 1, p(one_l), p(one_m), p(frames[i]), p(zero), 200, p(op), p(zero), p(onf))
        dt = time.perf_counter() - t
        if dt > 5e-4: slow.append((i, round(dt * 1e6), int(onf[0]), rc))
    print("rep", rep, "slow calls:", slow)
