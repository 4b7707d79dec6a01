"""Relay launches of the 64-row kernel against plain ones, interleaved in one process (NOTES 10.6).
python3 tests/dev/relay_ab.py [pieces ...]   -- columns: KBEST_RELAY=0, the launch plan's own choice, then each forced count.
Medians over four interleaved rounds of the fastest of three timed launches (HIP events on the launch stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

dev = torch.device("cuda", 0)
forced = [int(a) for a in sys.argv[1:]]


def engine(**env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env:
        del os.environ[k_]
    return e


st = torch.cuda.Stream()
for cfg, B in (("c4", None), ("c3", None), ("c4", 128), ("c4", 700), ("c4", 768), ("c4", 1536), ("c4", 2048), ("c3", 2048), ("c3", 8192)):
    costs, N, M, k = wl.dense_config(cfg, B=B)
    B = costs.shape[0]
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    names = ["plain", "plan"] + [f"{p} pieces" for p in forced]
    engs = [engine(KBEST_RELAY=0), engine()] + [engine(KBEST_RELAY=p) for p in forced]
    res = {n: [] for n in names}
    sums = {}
    for rnd in range(4):
        for n, e in zip(names, engs):
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[n].append(min(ts[1:]))
            sums[n] = (d_g.sum().item(), int(d_n.sum().item()), int(d_r.sum().item()))
    assert len(set(sums.values())) == 1, sums
    print(f"{cfg} B={B}: " + "  ".join(f"{n}: {np.median(res[n]):.3f}" for n in names), flush=True)
