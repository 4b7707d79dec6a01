"""Dev aid (GPU box): long runs of exactly equal gains -- the case ADVICE r5 named (all-equal costs at large k): time of the finishing
launch's ordering (radix over the columns since round 6; L^2 lexicographic comparisons before) and the canonical order checked."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import probabilisticsemslam_amd as pk
eng = pk.KBestEngine(0)
dev = torch.device("cuda", 0)
for (N, M, k, B, i8) in ((64, 64, 1000, 4, False), (64, 64, 1000, 4, True), (16, 16, 4000, 2, False), (30, 10, 3000, 3, False), (64, 64, 200, 64, False)):
    costs = np.zeros((B, N * M))
    tdt = torch.int8 if i8 else torch.int32
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.empty((B, k, M), dtype=tdt, device=dev); d_c = torch.empty((B, k, N), dtype=tdt, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev); d_n = torch.empty(B, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    ts = []
    for tc in (True, False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=s.cuda_stream, d_tie_flags=d_f if tc else None, tables_i8=i8, tie_check=tc)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    r = d_r.cpu().numpy().astype(np.int64); nf = d_n.cpu().numpy(); fl = d_f.cpu().numpy()
    ok = True
    for b in range(B):
        rows = [tuple(x) for x in r[b, : nf[b]]]
        ok = ok and rows == sorted(rows) and len(set(rows)) == len(rows)
        c = d_c.cpu().numpy().astype(np.int64)[b, : nf[b]]
        for sidx in range(0, int(nf[b]), max(1, int(nf[b]) // 7)):
            inv = np.full(N, -1); inv[r[b, sidx]] = np.arange(M)
            got = c[sidx].copy(); got[got >= M] = -1
            ok = ok and (got == inv).all()
    print(f"{B} x {N}x{M} all-zero costs, k={k}, int8={i8}: with the ordering {ts[2]:.2f} ms, without (KBEST_FLAG_NO_TIE_CHECK) {ts[1]:.2f} ms; nf {nf[:3]} flags {[hex(int(x)) for x in fl[:3]]}; lexicographic and distinct: {ok}")
