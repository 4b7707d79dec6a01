"""A few registered-buffer calls of kbest_batch_f64 on C4 (for a rocprofv3 timeline).  Development aid."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
B, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
costs = np.ascontiguousarray(wl.dense_batch(B, N, M, seed))
eng = pk.KBestEngine(0)
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)
o = eng._opts(False, None)
p = lambda a: a.ctypes.data_as(C.c_void_p)
eng.register_host(costs, r4c, c4r, gain, nf)
for i in range(4):
    t0 = time.perf_counter()
    rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), None, p(gain), p(nf), None)
    print("call %d: %.2f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)

