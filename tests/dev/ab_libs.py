"""A/B of in-tree builds in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): kernel time of a dense config.
usage: python tests/dev/ab_libs.py c4|c3|c2|NxM:k:B [B] libA.so libB.so ...      (library names relative to probabilisticsemslam_amd/)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import engine as E, workloads as wl

args = sys.argv[1:]
cfg = args.pop(0)
B = int(args.pop(0)) if args and args[0].isdigit() else None
libs = args
if "x" in cfg:  # custom shape "NxM:k:B"
    shp, k, Bc = cfg.split(":")
    N, M = (int(x) for x in shp.split("x")); k = int(k); Bc = int(Bc); seed = 0x5EED0000 + 1000 * N + k
else:
    Bc, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
B = B or Bc
costs = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
ts = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ts); s = ts.cuda_stream
engs = []
for lib in libs:
    os.environ["KBEST_LIB"] = lib
    E._lib = None
    e = pk.KBestEngine(0); e.reserve(B, N, k)
    e.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s); torch.cuda.synchronize()
    engs.append((lib, e, g.sum().item()))
times = {lib: [] for lib in libs}
for rnd in range(12):
    for lib, e, _ in engs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): e.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf, stream=s)
        e1.record(); torch.cuda.synchronize()
        times[lib].append(e0.elapsed_time(e1) / 4)
for lib, e, gs in engs:
    t = np.array(times[lib][2:])
    print(f"{cfg} B={B} {lib:28s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  gsum {gs:.9e}", flush=True)
