"""Dev aid (GPU box): latency of ONE problem on the reference-order kernel (KBEST_FLAG_REFERENCE_ORDER) next to the default route and,
for a problem with exact ties, KBEST_FLAG_REFERENCE_TIES -- what a per-frame caller pays.  Host entry, median of 20 calls."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk

eng = pk.KBestEngine(0)
rng = np.random.default_rng(2)
for (N, M, k, integer) in ((6, 3, 20, True), (28, 10, 200, True), (28, 10, 200, False), (64, 64, 200, False)):
    C = rng.integers(0, 12, (1, N * M)).astype(float) if integer else rng.random((1, N * M))
    row = []
    for kw in ({"canonical_ties": True}, {}, {"reference_order": True}):
        ts = []
        for i in range(22):
            t = time.perf_counter()
            out = eng.kbest(C, N, M, k, tie_flags=True, **kw)
            ts.append(time.perf_counter() - t)
        row.append("%8.1f us" % (1e6 * float(np.median(ts[2:]))))
    print(N, M, k, "integer" if integer else "continuous", "canonical_ties / default (reference ties) / reference_order:", *row, "flags", hex(int(out[-1][0])))
