"""Dev aid (GPU box, PROFILE build): which frames of the 1 000-frame C5 launch decide its length -- passes, overflows, cycles."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("KBEST_LIB", "libkbest_amd_prof.so")
import numpy as np, torch
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
F = 1000
fr = wl.kitti_like_frames(F)
prof = torch.zeros((F, 16), dtype=torch.int64, device="cuda")
eng.lib.kbest_set_profile_buffer(eng.ctx, C.c_void_p(prof.data_ptr()))
eng.register_host(*[])  # (nothing: host path)
out, nf = eng.weights(fr, [20] * F, [10] * F, 200, condition=True)
out, nf = eng.weights(fr, [20] * F, [10] * F, 200, condition=True)
p = prof.cpu().numpy()
passes, over, cyc = p[:, 0], p[:, 1], p[:, 5]
work = cyc - p[:, 2]   # without the load over PCIe
print("cycles without the PCIe load: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (work.mean(), *np.percentile(work, [50, 90, 99]), work.max()))
for n in np.unique(passes):
    m = passes == n
    print(f"  {int(n)} passes: {m.sum():4d} frames, of which with a lowered / abandoned pass {int((over[m] > 0).sum()):4d}; cycles mean {work[m].mean():.0f} max {work[m].max():.0f}; counting {p[m, 11].mean():.0f}, collecting {p[m, 12].mean():.0f}")
top = np.argsort(-work)[:12]
print("the slowest frames:", [(int(i), int(passes[i]), int(over[i]), int(work[i])) for i in top])
