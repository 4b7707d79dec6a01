"""Per-phase wave-cycle split of the general-size kernel (PROFILE build: make -C probabilisticsemslam_amd/csrc PROFILE=1).
Run on the GPU box: python tests/dev/wide_phases.py N M k B.  Development aid."""
import os, sys, ctypes as C
import numpy as np
os.environ.setdefault("KBEST_LIB", "libkbest_amd_prof.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import probabilisticsemslam_amd as pk
N, M, k, B = (int(x) for x in sys.argv[1:5])
if N <= 64: os.environ["KBEST_FORCE_WIDE"] = "1"
eng = pk.KBestEngine(0)
rng = np.random.default_rng(0)
costs = torch.from_numpy(rng.random((B, N * M)) * 50).to(dev)
r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev); c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
g = torch.empty((B, k), dtype=torch.float64, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev)
prof = torch.zeros((B, 16), dtype=torch.int64, device=dev)
eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf); torch.cuda.synchronize()
prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.kbest_dev(costs, B, N, M, k, r4c, c4r, g, nf); e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64).mean(axis=0)
names = ["cost copy", "root", "(rounds)", "select+emit bookkeeping", "outputs", "ticket+state load", "dijkstra", "update+augment", "gain", "store+push",
         "barrier wait", "(children started)", "(children completed)", "-", "merge", "whole"]
tot = p[15]
print(f"{N}x{M} k={k} B={B}: {e0.elapsed_time(e1):.2f} ms; per problem: rounds {p[2]/ (p[15] and 1):.0f} (summed over waves), children {p[11]:.0f}, completed {p[12]:.0f}, steps {p[13]:.0f}")
for i, n in enumerate(names):
    if n.startswith("(") or n == "-": continue
    print(f"  {n:28s} {p[i]:14.0f} ticks  {100*p[i]/tot:5.1f} %")
if p[11]: print(f"  per child: load {p[5]/p[11]:.0f}, dijkstra {p[6]/p[11]:.0f} ({p[6]/max(p[13],1):.0f} per step); per completed: update {p[7]/p[12]:.0f}, gain {p[8]/p[12]:.0f}, store {p[9]/p[12]:.0f} ticks")
