"""kbest_batch_f64 on C4 with registered caller buffers: int32 tables against int8 tables (KBEST_FLAG_TABLES_I8).  Development aid."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.zeros(1, device="cuda")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import engine as E, workloads as wl
B, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
costs = np.ascontiguousarray(wl.dense_batch(B, N, M, seed))
eng = pk.KBestEngine(0)
gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)
p = lambda a: a.ctypes.data_as(C.c_void_p)
eng.register_host(costs, gain, nf)
for name, dt, fl in (("int32", np.int32, 0), ("int8", np.int8, E.KBEST_FLAG_TABLES_I8), ("int32", np.int32, 0), ("int8", np.int8, E.KBEST_FLAG_TABLES_I8)):
    r4c = np.zeros((B, k, M), dt); c4r = np.zeros((B, k, N), dt)
    o = eng._opts(False, None, fl)
    for reg in (False, True):
        if reg: eng.register_host(r4c, c4r)
        for w in (True, False):
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r) if w else None, p(gain), p(nf), None)
                assert rc == 0
                ts.append(1e3 * (time.perf_counter() - t0))
            print(f"{name} registered={reg} col4row={w}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    eng.unregister_host(r4c, c4r)
