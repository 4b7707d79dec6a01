"""kbest_assoc_probs_batch_f64 with F C5 frames in one call: host-inclusive ms.  Development aid (KBEST_ZC_LIMIT_KB, KBEST_NO_POLL)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.zeros(1, device="cuda")
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
k, nL, nM = 200, 20, 10
eng = pk.KBestEngine(0)
p = lambda a: a.ctypes.data_as(C.c_void_p)
for F in [int(a) for a in sys.argv[1:]] or [1000]:
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM)
    nR = nL + nM
    raw = np.ascontiguousarray(np.concatenate(frames))
    h_nL, h_nM = np.full(F, nL, np.int32), np.full(F, nM, np.int32)
    h_coff = np.arange(F, dtype=np.int64) * nR * nM
    h_poff = np.arange(F, dtype=np.int64) * nM * (nL + 1)
    hp, hnf = np.zeros(F * nM * (nL + 1)), np.zeros(F, np.int32)
    ts = []
    for _ in range(8):
        t1 = time.perf_counter()
        rc = eng.lib.kbest_assoc_probs_batch_f64(eng.ctx, F, p(h_nL), p(h_nM), p(raw), p(h_coff), k, p(hp), p(h_poff), p(hnf))
        ts.append(1e3 * (time.perf_counter() - t1))
        assert rc == 0
    print(f"F={F}: " + " ".join(f"{t:.3f}" for t in ts) + f"  sum nf {int(hnf.sum())} checksum {hp.sum():.9f}", flush=True)
    ref = hp.copy(); hp[:] = 0
    eng.register_host(raw, hp)
    ts = []
    for _ in range(8):
        t1 = time.perf_counter()
        rc = eng.lib.kbest_assoc_probs_batch_f64(eng.ctx, F, p(h_nL), p(h_nM), p(raw), p(h_coff), k, p(hp), p(h_poff), p(hnf))
        ts.append(1e3 * (time.perf_counter() - t1))
        assert rc == 0
    eng.unregister_host(raw, hp)
    print(f"F={F} registered: " + " ".join(f"{t:.3f}" for t in ts) + f"  same {np.array_equal(ref, hp)}", flush=True)
