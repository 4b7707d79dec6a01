"""The fused association launch (C5's kernel) against the number of frames per call: where does a second generation start?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
dev = torch.device("cuda", 0)
eng = pk.KBestEngine(0)
st = torch.cuda.Stream()
k, nL, nM = 200, 20, 10
nR = nL + nM
for F in [int(a) for a in sys.argv[1:]] or [500, 900, 1000, 1024, 1030, 1100, 1300, 1536, 2000, 2048, 3000, 4000]:
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM)
    raw = np.ascontiguousarray(np.concatenate(frames))
    d_cost = torch.from_numpy(raw).to(dev)
    d_nL = torch.full((F,), nL, dtype=torch.int32, device=dev); d_nM = torch.full((F,), nM, dtype=torch.int32, device=dev)
    d_nRow = torch.full((F,), nR, dtype=torch.int32, device=dev)
    d_coff = torch.arange(F, dtype=torch.int64, device=dev) * nR * nM; d_poff = torch.arange(F, dtype=torch.int64, device=dev) * nM * (nL + 1)
    d_probs = torch.zeros(F * nM * (nL + 1), dtype=torch.float64, device=dev); d_nf = torch.zeros(F, dtype=torch.int32, device=dev)
    eng.reserve_assoc(F, nR, nM, k)
    torch.cuda.synchronize()
    ts = []
    for it in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            a.record(); eng.assoc_probs_dev(F, nR, nM, d_nL, d_nM, d_nRow, d_cost, d_coff, k, d_probs, d_poff, d_nf, stream=st.cuda_stream); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = min(ts[2:])
    print(f"{F:5d} frames: {t:.3f} ms  = {1e3 * t / F:.3f} us per frame", flush=True)
