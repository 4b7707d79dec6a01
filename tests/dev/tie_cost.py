"""Dev aid (GPU box): what the DEFAULT rule's completion of tied gain levels costs on a batch where nearly every problem has one
(1 000 integer 28x10 problems, k = 200): the first pass alone, the whole call, and all 1 000 problems at each step's k (k + 64, 256,
1 024, 4 096) with the number of levels that close inside that table.  The numbers of NOTES 11.9 (d)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
E = pk.engine
eng = pk.KBestEngine(0)
rng = np.random.default_rng(3)
costs = rng.integers(0, 12, size=(1000, 280)).astype(np.float64)
N, M, k = 28, 10, 200
def t(fn, n=3):
    ts=[]
    for i in range(n):
        a=time.perf_counter(); r=fn(); ts.append(1e3*(time.perf_counter()-a))
    return min(ts), r
ms, r = t(lambda: eng.kbest(costs, N, M, k, tie_flags=True, tie_resolve=False))
fl = r[-1]
print("first pass only %.1f ms; boundary-flagged %d" % (ms, int(((fl & E.KBEST_TIE_BOUNDARY)!=0).sum())))
ms, r = t(lambda: eng.kbest(costs, N, M, k, tie_flags=True))
fl = r[-1]
print("with completion %.1f ms; resolved %d unresolved %d" % (ms, int(((fl & E.KBEST_TIE_RESOLVED)!=0).sum()), int(((fl & E.KBEST_TIE_UNRESOLVED)!=0).sum())))
for k2 in (264, 456, 1224, 4296):
    ms, r = t(lambda: eng.kbest(costs, N, M, k2, tie_flags=True, tie_resolve=False), 2)
    g = r[3]; nf = r[0]
    closed = int(sum(1 for b in range(1000) if nf[b] < k2 or g[b, k2-1] != g[b, k-1]))
    print("all 1000 at k = %d: %.1f ms (route %s); level at slot %d closed inside for %d" % (k2, ms, hex(eng.last_route()), k, closed))
    ms2, _ = t(lambda: eng.kbest(costs, N, M, k2, tie_flags=True, tie_resolve=False, tables_i8=True), 2)
    print("   ... with int8 tables %.1f ms" % ms2)
