import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
costs, N, M, k = wl.dense_config("c4", B=1024)
eng.kbest(costs, N, M, k)
t0 = time.perf_counter()
for _ in range(5):
    nf = eng.kbest(costs, N, M, k)[0]
dt = (time.perf_counter() - t0) / 5
print(f"host-pointer kbest_batch_f64, 1024 x 64x64 k=200: {dt*1e3:.2f} ms per call, {nf.sum()/dt:.3e} assignments/s")
