"""kbest_batch_f64 on C4 with pageable / registered caller buffers, with and without col4row.  Development aid."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.zeros(1, device="cuda")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
B, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
costs = np.ascontiguousarray(wl.dense_batch(B, N, M, seed))
eng = pk.KBestEngine(0)
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)
o = eng._opts(False, None)
p = lambda a: a.ctypes.data_as(C.c_void_p)
def call(with_c4r):
    t0 = time.perf_counter()
    rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r) if with_c4r else None, p(gain), p(nf), None)
    assert rc == 0
    return 1e3 * (time.perf_counter() - t0)
for name in ("pageable", "registered"):
    if name == "registered": eng.register_host(costs, r4c, c4r, gain, nf)
    for w in (True, False, True, False):
        ts = [call(w) for _ in range(6)]
        print(f"{name} col4row={w}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
eng.unregister_host(costs, r4c, c4r, gain, nf)
