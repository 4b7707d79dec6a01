"""Dev loop (GPU box): per-frame diagnostics of the bounded-walk kernel (passes, partial assignments visited, abandoned passes)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
for F in (1, 8, 300, 1000):
    fr = wl.kitti_like_frames(F)
    prof = torch.zeros((F, 16), dtype=torch.int64, device="cuda")
    eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
    out, nf = eng.weights(fr, [20] * F, [10] * F, 200, condition=True)
    p = prof.cpu().numpy()
    print(f"F={F}: passes mean {p[:,0].mean():.1f} max {p[:,0].max()}, passes that did not fit {p[:,1].sum()}, candidates mean {p[:,3].mean():.0f} max {p[:,3].max()}, cycles mean {p[:,5].mean():.0f} max {p[:,5].max()}, nf min {nf.min()}")

    print("   cycles: set-up %.0f, counting passes %.0f, collecting walk %.0f, sort + weights %.0f" % tuple(p[:, 10:14].mean(axis=0)))
    print("   set-up cycles: load %.0f, condition %.0f, greedy %.0f, column sort %.0f, order + bound tables %.0f" % (p[:, 2].mean(), p[:, 4].mean(), p[:, 14].mean(), p[:, 15].mean(), (p[:, 10] - p[:, 2] - p[:, 4] - p[:, 14] - p[:, 15]).mean()))
    print("   sort + weights cycles: rank sort %.0f, scatter %.0f, exp %.0f, accumulate + write %.0f" % tuple(p[:, 6:10].mean(axis=0)))
    bad = np.nonzero(nf == -2)[0]
    for i in bad[:6]:
        print("   handed back: frame", int(i), "passes", int(p[i, 0]), "did not fit", int(p[i, 1]), "last count", int(p[i, 7]), "Ulo", p[i, 8:9].view(np.float64)[0], "Uhi", p[i, 9:10].view(np.float64)[0])
