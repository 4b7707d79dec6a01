"""Kernel time of the general-size kernel (kbest_wide.hip) with device-resident buffers.  Development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
rng = np.random.default_rng(0)
for N, M, k, B, force in ((128, 128, 200, 256, 0), (128, 128, 200, 512, 0), (96, 96, 200, 512, 0), (256, 256, 200, 256, 0), (100, 20, 200, 1024, 0), (64, 64, 200, 256, 1)):
    if force:
        os.environ["KBEST_FORCE_WIDE"] = "1"
    eng = pk.KBestEngine(0)
    costs = torch.from_numpy(rng.random((B, N * M)) * 50).to(dev)
    r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    s = ts.cuda_stream
    eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{N}x{M} k={k} B={B}{' (forced wide)' if force else ''}: {ms:.2f} ms kernel, {float(nf.sum())/ms*1e3:.3e} assignments/s", flush=True)
