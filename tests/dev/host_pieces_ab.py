"""Dev aid (GPU box): the host entry (1 024 x 64x64, k = 200, pageable and registered buffers) under KBEST_PIECES = 2, 3, 4 -- one
context per setting (the knob is read when a context is created), calls interleaved over the settings, median / min of 15 calls each."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

_, N, M, k, seed = wl.DENSE_CONFIGS["c4"]
B = 1024
costs = wl.dense_batch(B, N, M, seed)
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); gain = np.zeros((B, k)); nf = np.zeros(B, np.int32)
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
engs = {}
for P in (2, 3, 4):
    os.environ["KBEST_PIECES"] = str(P)
    engs[P] = pk.KBestEngine(0)
o = pk.engine.KBestOpts()
engs[2].lib.kbest_default_opts(C.byref(o))
for leg in ("pageable", "registered"):
    if leg == "registered":
        for a in (costs, r4c, c4r, gain, nf):
            assert engs[2].lib.kbest_register_host_buffer(engs[2].ctx, p(a), C.c_size_t(a.nbytes)) == 0  # (pinning is the process's: every context sees it)
    ts = {P: [] for P in engs}
    for i in range(17):
        for P, e in engs.items():
            t = time.perf_counter()
            assert e.lib.kbest_batch_f64(e.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(gain), p(nf), None) == 0
            ts[P].append(1e3 * (time.perf_counter() - t))
    print(leg, {P: "median %.2f min %.2f" % (np.median(v[2:]), min(v[2:])) for P, v in ts.items()})
