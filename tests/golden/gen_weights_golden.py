"""Generate tests/golden/weights_golden.npz from the reference's OWN weights functions.

Run in the build container only (needs /root/reference):

    make -C oracle && python tests/golden/gen_weights_golden.py

It loads oracle/_ref/libref_assign.so -- verbatim line ranges of /root/reference/assignment.cpp
(conditionCosts :439-525, toProbs :527-542, assignmentProb :547-683, bruteForceProb :835-964; see
oracle/ref_assign_shim.cpp and oracle/Makefile for how they are cut and compiled, -O2 strict IEEE)
linked against the unmodified solver -- and records, for seeded cost blocks, exactly what those
functions return.  The fixture holds data only (inputs and expected outputs), no reference source.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as ol  # noqa: E402
from probabilisticsemslam_amd import workloads as wl  # noqa: E402

INF = float("inf")


def frames():
    """Yields (name, raw cost block (nL+nM) x nM col-major, nL, nM, k, brute)."""
    for i, f in enumerate(wl.kitti_like_frames(6)):                      # C5 frames, seed 0xC0FFEE
        yield f"c5_f{i}", f, 20, 10, 200, False
    for i, f in enumerate(wl.kitti_like_frames(6, nL=6, nM=3)):          # small exhaustive frames
        yield f"small_f{i}", f, 6, 3, 200, True
    for i, f in enumerate(wl.kitti_like_frames(3, nL=12, nM=5, seed=0xBEEF01)):
        yield f"mid_f{i}", f, 12, 5, 100, True
    for i, f in enumerate(wl.kitti_like_frames(2, nL=40, nM=12, seed=0xBEEF02)):
        yield f"wide_f{i}", f, 40, 12, 200, False
    for i, f in enumerate(wl.kitti_like_frames(3, nL=9, nM=1, seed=0xBEEF03)):   # single-column fast path (:554-570)
        yield f"onecol_f{i}", f, 9, 1, 50, True
    for i, f in enumerate(wl.kitti_like_frames(2, nL=3, nM=6, seed=0xBEEF04)):   # more measurements than landmarks
        yield f"fewland_f{i}", f, 3, 6, 300, True
    for i, kk in enumerate((1, 20, 100, 1000)):                          # the harness's k sweep (comparison.cpp:194-243)
        yield f"ksweep_k{kk}", wl.kitti_like_frames(1, seed=0xBEEF05)[0], 20, 10, kk, False
    # no landmark within the gate: everything goes to the dummy rows
    f = np.full(13 * 3, INF)
    f[:] = INF
    for c in range(3):
        for r in range(10):
            f[c * 13 + r] = 300.0 + 7 * r + c
        f[c * 13 + 10 + c] = 10.0
    yield "all_gated", f, 10, 3, 50, True


def main():
    lib = ol.ref_assign()
    out, names = {}, []
    for name, raw, nL, nM, k, brute in frames():
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        nR = nL + nM
        cond, ridx = ol.ref_condition_costs(raw, nR, nM)
        condL = len(ridx) - nM
        p = ol.ref_assignment_prob(cond, condL, nM, k)
        names.append(name)
        out[name + "/raw"] = raw
        out[name + "/meta"] = np.array([nL, nM, k, len(ridx), int(brute)], dtype=np.int64)
        out[name + "/cond"] = cond
        out[name + "/rowIdx"] = ridx.astype(np.int16)
        out[name + "/probs"] = p
        tp = cond.copy()
        lib.ref_to_probs(tp, tp.size)
        out[name + "/toProbs"] = tp
        if brute:
            out[name + "/brute"] = ol.ref_brute_force_prob(cond, condL, nM)
        print(f"{name:14s} nL={nL:3d} nM={nM:3d} k={k:4d} kept rows={len(ridx):3d} brute={int(brute)} p00={p.reshape(-1)[0]:.6g}")
    out["names"] = np.array(names)
    path = os.path.join(HERE, "weights_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
