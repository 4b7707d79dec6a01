"""One matrix over several workgroups (split_factor, kbest_capi.cpp): parity with the oracle and kernel time against the
unsplit launch, for small batches of 64x64 / 48x48 problems.  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol

def engine(**env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    e = pk.KBestEngine(0)
    for k_ in env: del os.environ[k_]
    return e

engs = {"nosplit": engine(KBEST_NO_SPLIT=1), "auto": engine(), "split2": engine(KBEST_SPLIT=2)}
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
for (N, k) in ((64, 200), (48, 100), (64, 50)):
    for B in (1, 8, 32, 64, 128):
        costs = wl.dense_batch(B, N, N, 0x5EED0000 + 1000 * N + k)
        d_cost = torch.from_numpy(costs).to(dev)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs[: min(B, 8)], N, N, k)
        line = f"{N}x{N} k={k} B={B:3d}:"
        for name, e in engs.items():
            r4c = torch.empty((B, k, N), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
            g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
            e.reserve(B, N, k)
            e.kbest_dev(d_cost, B, N, N, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
            nb = min(B, 8)
            ok = (nf[:nb].cpu().numpy() == onf).all() and (r4c[:nb].cpu().numpy() == or4c).all() and \
                 (g[:nb].cpu().numpy().view(np.int64) == og.view(np.int64)).all() and (c4r[:nb].cpu().numpy() == oc4r).all()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): e.kbest_dev(d_cost, B, N, N, k, r4c, c4r, g, nf, stream=s)
            e1.record(); torch.cuda.synchronize()
            line += f"  {name} {e0.elapsed_time(e1)/5:.3f} ms{'' if ok else ' MISMATCH'}"
        print(line, flush=True)
