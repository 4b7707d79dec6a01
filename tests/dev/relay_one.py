import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
dev = torch.device("cuda", 0)
cfg = sys.argv[1]
Bq = int(sys.argv[2]) if len(sys.argv) > 2 else None
costs, N, M, k = wl.dense_config(cfg, B=Bq)
B = costs.shape[0]
d_cost = torch.from_numpy(costs).to(dev)
d_r = torch.empty((B, k, N), dtype=torch.int32, device=dev); d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
e = pk.KBestEngine(0)
st = torch.cuda.Stream()
ts = []
for it in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        a.record(); e.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st.cuda_stream); b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(cfg, B, os.environ.get("KBEST_RELAY"), os.environ.get("KBEST_NWAVES"), "min %.3f" % min(ts[1:]), "gsum %.9e" % d_g.sum().item(), "nf", int(d_n.sum().item()), flush=True)
