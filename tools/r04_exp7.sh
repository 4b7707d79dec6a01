#!/bin/bash
# tools/r04_exp7.sh -- on the GPU box: narrow staging of the host entry: tests, then the host-inclusive numbers by host thread count
out=$(pwd)/gpurun_out/r04_exp7
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "narrow or ragged" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
nproc
for t in 8 16 0 "0 KBEST_ZC_COST=0"; do
  env KBEST_HOST_THREADS=$t timeout 300 python3 - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
costs, N, M, k = wl.dense_config("c4")
B = costs.shape[0]
eng = pk.KBestEngine(0)
r4c = np.zeros((B, k, M), np.int32); c4r = np.zeros((B, k, N), np.int32); g = np.zeros((B, k)); nf = np.zeros(B, np.int32)
o = eng._opts(False, None)
p = lambda a: a.ctypes.data_as(C.c_void_p)
def timed(n=6):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(g), p(nf), None)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    return 1e3 * best
pg = timed()
eng.register_host(costs)
rg = timed()
print(f"host threads {os.environ['KBEST_HOST_THREADS']} zc {os.environ.get('KBEST_ZC_COST','1')}: pageable {pg:.3f} ms, cost blocks registered {rg:.3f} ms")
PY
done
