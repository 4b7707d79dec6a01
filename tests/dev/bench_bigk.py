"""k in the thousands (bruteForceProb's call pattern, assignment.cpp:858-880): general-size kernel with the pool in HBM.
Development aid: kernel time per batch vs the reference solver on one core."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
import oracle_lib as ol
eng = pk.KBestEngine(0)
rng = np.random.default_rng(5)
for N, M, k, B in ((12, 12, 5000, 256), (30, 10, 20000, 64), (30, 10, 20000, 256), (16, 16, 20000, 256), (64, 64, 2000, 256)):
    costs_h = rng.random((B, N * M)) * 20
    costs = torch.from_numpy(costs_h).to(dev)
    r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
    eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t0 = time.perf_counter()
    nref = 2
    for b in range(nref):
        rr = ol.ref_kbest(costs_h[b], N, M, k, ofast=True)
    tref = (time.perf_counter() - t0) / nref * 1e3
    print(f"{N}x{M} k={k} B={B}: {ms:.2f} ms per batch = {ms/B*1e3:.0f} us per problem, nf mean {float(nf.float().mean()):.0f}; reference {tref:.1f} ms per problem", flush=True)
