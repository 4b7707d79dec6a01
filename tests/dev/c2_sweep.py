"""dev aid: the lane-per-child kernel under KBEST_LANE_SPEC (hypotheses split per round) over batch shapes, interleaved on one box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
dev = torch.device("cuda", 0)
def engine(**env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env: del os.environ[k_]
    return e
specs = [8, 10, 12, 14, 16]
engs = [engine(KBEST_LANE_SPEC=s) for s in specs]
st = torch.cuda.Stream()
for (B, N, k) in ((1024, 16, 50), (600, 16, 50), (2048, 16, 50), (4096, 16, 50), (1024, 16, 200), (1024, 12, 50), (1024, 16, 10), (8192, 16, 200), (300, 16, 50)):
    costs = wl.dense_batch(B, N, N, 0x5EED0000 + 1000 * N + k)
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    res = {s: [] for s in specs}
    for rnd in range(3):
        for s, e in zip(specs, engs):
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, N, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[s].append(min(ts[1:]))
    print(f"{B:5d} x {N}x{N}, k = {k:3d}: " + "  ".join(f"spec {s}: {np.median(res[s]):.4f}" for s in specs), flush=True)
