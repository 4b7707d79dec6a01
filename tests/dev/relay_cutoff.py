"""Relay plan against plain launches where a cutoff ends most matrices early (their later pieces' workgroups start, find the matrix
finished and leave): is the plan ever slower?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk

dev = torch.device("cuda", 0)


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env:
        del os.environ[k_]
    return e


st = torch.cuda.Stream()
rng = np.random.default_rng(6)
for (N, M, k, B, cutoff) in ((64, 64, 200, 2048, 0.05), (64, 64, 200, 2048, 0.02), (64, 64, 200, 2048, 0.005), (32, 32, 200, 6000, 0.02), (32, 32, 200, 6000, 0.002),
                             (40, 12, 200, 3000, 0.5)):
    costs = rng.random((B, N * M))
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, M), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    names = ["plain", "plan"]
    engs = [engine(KBEST_RELAY=0, KBEST_NO_SMALL=1, KBEST_NO_LANE=1), engine(KBEST_NO_SMALL=1, KBEST_NO_LANE=1)]
    res = {n: [] for n in names}
    sums = {}
    for rnd in range(3):
        for n, e in zip(names, engs):
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream, cutoff=cutoff); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[n].append(min(ts[1:]))
            sums[n] = int(d_n.sum().item())
    assert len(set(sums.values())) == 1, sums
    print(f"{N}x{M} k={k} B={B} cutoff={cutoff}: " + "  ".join(f"{n}: {np.median(res[n]):.3f}" for n in names) + f"   (mean nf {sums['plain'] / B:.1f}; relays {engs[1].relay_launches()})", flush=True)
