"""dev aid: optimistic bounds on / off (KBEST_NO_OPT) on dense square batches of other sizes than C3's"""
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
on, off = engine(), engine(KBEST_NO_OPT=1)
st = torch.cuda.Stream()
for (B, N, k) in ((4096, 32, 50), (4096, 32, 20), (8192, 24, 200), (4096, 28, 100), (16384, 20, 50), (4096, 32, 400), (4096, 17, 200)):
    costs = wl.dense_batch(B, N, N, 0xAB0000 + N * 1000 + k)
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    res = {}
    for name, e in (("on", on), ("off", off)):
        ts = []
        for it in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                a.record(); e.kbest_dev(d_cost, B, N, N, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[name] = (min(ts[1:]), d_g.cpu().numpy().copy())
    same = (res["on"][1].view(np.int64) == res["off"][1].view(np.int64)).all()
    print(f"{B} x {N}x{N}, k = {k}: off {res['off'][0]:.3f} ms, on {res['on'][0]:.3f} ms ({100*(res['on'][0]/res['off'][0]-1):+.1f} %), same gains {same}", flush=True)
