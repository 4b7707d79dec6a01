"""The reference's comparison harness (compMethods, comparison.cpp:62-341) on this engine.

For every saved per-frame cost matrix: read the .dat file, conditionCosts, "truth" = bruteForceProb (checker),
then assignmentProb for k in {1, 20, 100, 200, 1000} on the GPU (all frames of one k in ONE batched call), maximum
absolute probability error per frame (comparison.cpp:261-275), the reference's acceptance rule (abort above 0.1,
report above 1e-8, comparison.cpp:319-331) and the error order statistics its errOrdStats figure plots.  The
reference ships no cost matrices (they come out of a KITTI run), so the frames are written first by the synthetic
KITTI-like generator, through the reference's own file format.

    python tests/harness_comp_methods.py [n_frames]
"""
from __future__ import annotations

import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as ol  # noqa: E402
import probabilisticsemslam_amd as pk  # noqa: E402
from probabilisticsemslam_amd import costfile, workloads as wl  # noqa: E402

KS = (1, 20, 100, 200, 1000)


def run(n_frames: int = 64, nL: int = 6, nM: int = 3, directory: str | None = None, verbose: bool = True):
    eng = pk.KBestEngine(0)
    tmp = tempfile.TemporaryDirectory() if directory is None else None
    d = directory or tmp.name
    frames = wl.kitti_like_frames(n_frames, nL=nL, nM=nM)
    for i, f in enumerate(frames):  # semslamRun side: one file per frame (system.cpp:271)
        costfile.write_cost_matrix(os.path.join(d, f"synthetic_frame{i + 1}.dat"), f, nL + nM, nM)
    conds, condLs, truth = [], [], []
    for i in range(n_frames):       # compMethods side
        cost, fl, fm = costfile.read_cost_matrix(os.path.join(d, f"synthetic_frame{i + 1}.dat"))
        assert (fl, fm) == (nL, nM)
        cond, idx = ol.condition_costs(cost, fl + fm, fm)
        conds.append(cond)
        condLs.append(len(idx) - fm)
        truth.append(ol.brute_force_prob(cond, condLs[-1], fm)[0])
    table = {}
    for k in KS:
        t0 = time.perf_counter()
        probs, nf = eng.weights(conds, condLs, [nM] * n_frames, k)
        dt = time.perf_counter() - t0
        err = np.array([np.abs(p - t).max() for p, t in zip(probs, truth)])
        table[k] = dict(ms_per_frame=1e3 * dt / n_frames, err=err)
        if verbose:
            q = np.quantile(err, [0.5, 0.95, 1.0])
            print(f"k={k:5d}  {1e3 * dt / n_frames:8.4f} ms/frame (batched, PCIe incl.)  max-abs-error: median {q[0]:.2e} "
                  f"p95 {q[1]:.2e} worst {q[2]:.2e}  frames > 1e-8: {(err > 1e-8).sum()}")
    if tmp:
        tmp.cleanup()
    return table


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
