"""Relay plan against plain launches on shapes other than the bench configs (random dense costs; rectangular; cutoff; small k)."""
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
rng = np.random.default_rng(5)
for (N, M, k, B, cutoff) in ((64, 64, 50, 1024, None), (64, 64, 20, 2048, None), (48, 48, 100, 2000, None), (40, 40, 200, 1500, None), (64, 32, 200, 1024, None),
                             (64, 64, 200, 1024, 0.15), (33, 33, 200, 3000, None), (56, 56, 256, 900, None), (64, 64, 700, 800, None), (24, 24, 200, 6000, None), (64, 64, 1000, 700, None), (32, 32, 600, 2500, None)):
    costs = rng.random((B, N * M))
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, M), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    names = ["plain", "plan"]
    engs = [engine(KBEST_RELAY=0), engine()]
    res = {n: [] for n in names}
    sums = {}
    kw = {} if cutoff is None else {"cutoff": cutoff}
    for rnd in range(3):
        for n, e in zip(names, engs):
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream, **kw); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[n].append(min(ts[1:]))
            nf = d_n.cpu().numpy()
            sums[n] = (int(nf.sum()), float(np.nansum(d_g.cpu().numpy()[np.arange(k)[None, :] < nf[:, None]])))
    assert len(set(sums.values())) == 1, sums
    print(f"{N}x{M} k={k} B={B} cutoff={cutoff}: " + "  ".join(f"{n}: {np.median(res[n]):.3f}" for n in names) + f"   (nf sum {sums['plain'][0]})", flush=True)
