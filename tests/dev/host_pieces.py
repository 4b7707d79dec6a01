"""Host-pointer entry (kbest_batch_f64: H2D of the costs, D2H of every table) on C4 for KBEST_PIECES = 1 / 2 / 4, relay plan on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env:
        del os.environ[k_]
    return e


import ctypes as C
costs, N, M, k = wl.dense_config(sys.argv[1] if len(sys.argv) > 1 else "c4")
B = costs.shape[0]
costs = np.ascontiguousarray(costs)
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)
p = lambda a: a.ctypes.data_as(C.c_void_p)
ref = None
for pieces in ((None,) if os.environ.get('HP_ONE') else (None, 1, 2, 4)):
    for relay in ((None,) if os.environ.get('HP_ONE') else (None, 0)):
        env = {}
        if pieces is not None:
            env["KBEST_PIECES"] = pieces
        if relay is not None:
            env["KBEST_RELAY"] = relay
        e = engine(**env)
        o = e._opts(False, None)
        ts = []
        for it in range(20):
            t0 = time.perf_counter()
            rc = e.lib.kbest_batch_f64(e.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(gain), p(nf), None)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        out = (r4c.copy(), c4r.copy(), gain.copy())
        if ref is None:
            ref = out
        else:
            assert all((a == b).all() for a, b in zip(out, ref))
        print(f"pieces {pieces} relay {relay}: min {1e3 * min(ts[2:]):.3f} ms  median {1e3 * np.median(ts[2:]):.3f}", flush=True)
