"""Timing of the general-size kernel (kbest_wide.hip) on shapes beyond the LDS kernel.  Development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk

eng = pk.KBestEngine(0)
rng = np.random.default_rng(0)
for N, M, k, B in ((128, 128, 200, 256), (128, 128, 200, 512), (96, 96, 200, 512), (256, 256, 200, 256), (100, 20, 200, 1024),
                   (512, 512, 50, 64), (8, 8, 20000, 64), (64, 64, 200, 256)):
    costs = rng.random((B, N * M)) * 50
    if (N, M) == (64, 64):
        os.environ["KBEST_FORCE_WIDE"] = "1"
    eng.kbest(costs[:2], N, M, k)
    t0 = time.perf_counter()
    nf = eng.kbest(costs, N, M, k)[0]
    dt = time.perf_counter() - t0
    print(f"{N}x{M} k={k} B={B}: {dt*1e3:.1f} ms host-to-host, {nf.sum()/dt:.3e} assignments/s", flush=True)
