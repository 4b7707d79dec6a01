"""General-size kernel: the batch as a queue against the fixed stride (KBEST_NO_WIDE_QUEUE), batches larger than the grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
dev = torch.device("cuda", 0)
def engine(**env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env: del os.environ[k_]
    return e
st = torch.cuda.Stream()
rng = np.random.default_rng(3)
for (N, M, k, B) in ((128, 128, 200, 512), (128, 128, 200, 1200), (128, 128, 200, 2304), (100, 100, 100, 3000), (200, 200, 50, 1000), (96, 40, 200, 2500)):
    costs = rng.random((B, N * M))
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, M), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    names = ["stride", "queue"]
    engs = [engine(KBEST_NO_WIDE_QUEUE=1), engine()]
    res = {n: [] for n in names}; sums = {}
    for rnd in range(2):
        for n, e in zip(names, engs):
            ts = []
            for it in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[n].append(min(ts[1:]))
            sums[n] = (int(d_n.sum().item()), float(d_g.sum().item()), int(d_r.sum().item()))
    assert len(set(sums.values())) == 1, sums
    print(f"{N}x{M} k={k} B={B}: " + "  ".join(f"{n}: {np.median(res[n]):.3f}" for n in names), flush=True)
