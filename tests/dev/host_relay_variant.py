"""Dev aid (GPU box): VERDICT r5 item 7's variant of the host entry -- byte tables to DEVICE memory under ONE relay launch, then copied
home -- against the shipped path (four pieces, a piece's kernel behind its own upload, tables leaving as they become final).  The
one-device multi entry keeps its tables on the device (it needs them for the exchange), so KBEST_PIECES=1 there IS that variant (one
piece, no sub-batch, tables in device memory: kbest_capi.cpp's relay plan applies).
Median / min of 11 calls of 1 024 x 64x64, k = 200, pageable host buffers.  usage: python tests/dev/host_relay_variant.py (run once per
setting of KBEST_PIECES; the knob is read when a context is created)"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

_, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
costs = wl.dense_batch(1024, N, M, seed)
import ctypes as C
multi = pk.KBestMulti([0])
eng = pk.KBestEngine(0)
B = 1024
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)  # reused: no page faults in the timed calls
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
o = pk.engine.KBestOpts()
eng.lib.kbest_default_opts(C.byref(o))


def call_multi():
    assert multi.lib.kbest_batch_f64_multi(multi.m, C.byref(o), B, N, M, None, None, p(costs), k, p(r4c), p(c4r), p(gain), p(nf)) == 0


def call_single():
    assert eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(gain), p(nf), None) == 0


for name, fn in (("multi entry, one device (tables stay on the device, copied home)", call_multi),
                 ("single-device host entry (tables leave over the link as they become final)", call_single)):
    ts = []
    for i in range(13):
        t = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t))
    ts = ts[2:]
    print(f"KBEST_PIECES={os.environ.get('KBEST_PIECES', '(4)')}: {name}: median {np.median(ts):.2f} ms, min {min(ts):.2f}, max {max(ts):.2f}")
multi.close()
