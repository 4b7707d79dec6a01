"""Dense square batches: kernel time with the lane-per-child kernel forced on / off, over sizes.  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

def engine(**env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env: del os.environ[k_]
    return e

def run(eng, B, N, k):
    costs = torch.from_numpy(wl.dense_batch(B, N, N, 77 + N + k)).to(dev)
    r4c = torch.empty((B, k, N), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
    eng.reserve(B, N, k)
    eng.kbest_dev(costs, B, N, N, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.kbest_dev(costs, B, N, N, k, r4c, c4r, g, nf, stream=s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5, g.sum().item(), int(nf.sum().item())

old = engine(KBEST_NO_LANE=1)
small = len(sys.argv) > 1 and sys.argv[1] == "small"
if small:
    lanes = {f"nw{nw}s{sp}": engine(KBEST_FORCE_LANE=1, KBEST_LANE_NW=nw, KBEST_LANE_SPEC=sp) for nw in (4,) for sp in (4, 8, 12, 16)}
else:
    lanes = {f"nw{nw}s{sp}": engine(KBEST_FORCE_LANE=1, KBEST_LANE_NW=nw, KBEST_LANE_SPEC=sp) for nw in (2, 4) for sp in (4, 8)}
for N in ((8, 16, 24, 32) if small else (6, 8, 12, 16, 20, 24, 32)):
    for k in (10, 50, 200):
        for B in ((1, 64, 256, 512) if small else (600, 1024, 4096, 16384)):
            t0, g0, n0 = run(old, B, N, k)
            line = f"{N:2d}x{N:<2d} k={k:<3d} B={B:<5d} old {t0:7.3f} ms |"
            for name, e in lanes.items():
                t, g, n = run(e, B, N, k)
                line += f" {name} {t:7.3f}{'' if (g, n) == (g0, n0) else ' !!DIFF'}"
            print(line, flush=True)
