"""Generate tests/golden/assign_golden.npz from the UNMODIFIED reference solver: assign2D (shortestPathCPP.hpp:144-149)
and shortestPathCPP (hpp:178-182) on rectangular, maximise and infeasible problems, with the dual variables the
reference leaves in the MurtyHyp.  Run in the build container only:  make -C oracle && python tests/golden/gen_assign_golden.py
The fixture holds data only (inputs and expected outputs)."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as ol  # noqa: E402
from probabilisticsemslam_amd import workloads as wl  # noqa: E402

INF = float("inf")


def cases():
    """(name, cost N*M col-major, N, M, maximize, shift, gainCols)"""
    yield "sq_8x8", wl.splitmix64_u01(12345, 64), 8, 8, False, True, 0
    yield "sq_8x8_max", wl.splitmix64_u01(12345, 64), 8, 8, True, True, 0
    yield "rect_12x5", wl.splitmix64_u01(201, 60) * 10 - 3, 12, 5, False, True, 0
    yield "rect_12x5_max", wl.splitmix64_u01(201, 60) * 10 - 3, 12, 5, True, True, 0
    yield "rect_40x7", wl.splitmix64_u01(202, 280), 40, 7, False, True, 0
    yield "rect_64x20", wl.splitmix64_u01(203, 1280), 64, 20, False, True, 0
    yield "sq_64x64", wl.splitmix64_u01(204, 4096), 64, 64, False, True, 0
    yield "one_6x1", wl.splitmix64_u01(205, 6), 6, 1, False, True, 0
    yield "one_1x1", np.array([0.5]), 1, 1, False, True, 0
    c = wl.splitmix64_u01(206, 9 * 6)
    c = np.where(c < 0.35, -INF, c)
    for j in range(6):
        c[j * 9 + 3 + j] = 0.05
    yield "bb_like_9x6_max", c, 9, 6, True, True, 0          # asgnBB-style profits (assignment.cpp:724-797)
    f = wl.kitti_like_frames(1)[0]
    yield "kitti_30x10", f, 30, 10, False, True, 0
    c = wl.splitmix64_u01(207, 25).copy()
    c[10:15] = INF
    yield "infeasible_5x5", c, 5, 5, False, True, 0
    c = wl.splitmix64_u01(208, 6 * 4).copy()
    c[6:12] = INF
    yield "infeasible_6x4", c, 6, 4, False, True, 0
    # shortestPathCPP itself: non-negative matrix used as it is, gain over fewer columns (kBest2D calls it with
    # (D, D, numCol) on the padded matrix, cpp:587)
    yield "spc_10x10", wl.splitmix64_u01(209, 100), 10, 10, False, False, 0
    yield "spc_10x10_g4", wl.splitmix64_u01(209, 100), 10, 10, False, False, 4
    pad = np.concatenate([wl.splitmix64_u01(210, 12 * 5), np.zeros(12 * 7)])
    yield "spc_padded_12x12_g5", pad, 12, 12, False, False, 5
    yield "spc_rect_20x6", wl.splitmix64_u01(211, 120) * 7, 20, 6, False, False, 0
    c = wl.splitmix64_u01(212, 16).copy()
    c[4:8] = INF
    yield "spc_infeasible_4x4", c, 4, 4, False, False, 0


def main():
    out, names = {}, []
    for name, cost, N, M, maximize, shift, gc in cases():
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        ok, r4c, c4r, g, u, v = ol.ref_assign2d_ex(cost, N, M, maximize, shift, gc)
        names.append(name)
        out[name + "/cost"] = cost
        out[name + "/meta"] = np.array([N, M, int(maximize), int(shift), gc, ok], dtype=np.int64)
        out[name + "/row4col"] = r4c.astype(np.int16)
        out[name + "/col4row"] = c4r.astype(np.int16)
        out[name + "/gain"] = np.array([g])
        out[name + "/u"] = u
        out[name + "/v"] = v
        print(f"{name:22s} N={N:3d} M={M:3d} max={int(maximize)} shift={int(shift)} gainCols={gc} ok={ok} gain={g!r}")
    out["names"] = np.array(names)
    path = os.path.join(HERE, "assign_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
