"""dev aid: host-inclusive time of the multi-device entry on logical devices against the single-device host entry"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import engine as pe, workloads as wl
costs, N, M, k = wl.dense_config("c4")
B = costs.shape[0]
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); g = np.zeros((B, k)); nf = np.zeros(B, np.int32)
p = lambda a: a.ctypes.data_as(C.c_void_p)
eng = pk.KBestEngine(0)
o = eng._opts(False, None)
def t_single(n=6):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(g), p(nf), None)
        best = min(best, time.perf_counter() - t0); assert rc == 0
    return 1e3 * best
print(f"single-device host entry: {t_single():.3f} ms", flush=True)
for G in (1, 2, 8):
    m = pk.KBestMulti([0] * G)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        rc = m.lib.kbest_batch_f64_multi(m.m, C.byref(o), B, N, M, None, None, p(costs), k, p(r4c), p(c4r), p(g), p(nf))
        best = min(best, time.perf_counter() - t0); assert rc == 0
    print(f"multi entry, {G} logical devices ({os.environ.get('KBEST_NO_NARROW','narrow')}): {1e3*best:.3f} ms; timeline (ms) {np.round(1e3*m.timeline(), 2).tolist()}", flush=True)
    m.close()
