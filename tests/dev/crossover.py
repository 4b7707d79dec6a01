"""Where the GPU overtakes one host core: getAssignmentProbs per frame (conditionCosts -> assignmentProb(k=200), the reference's
own code, -Ofast, one thread) against kbest_assoc_probs_batch_f64 called with B frames at a time (host buffers in and out), for
the reference's real frame sizes ("3-5 measurements per frame", README.md:11) up to the C5 size.  Prints a table + JSON.
usage: python tests/dev/crossover.py [out.json]"""
import json, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
import oracle_lib as ol

k = 200
eng = pk.KBestEngine(0)
lib, ctx = eng.lib, eng.ctx
p = lambda a: a.ctypes.data_as(C.c_void_p)
rows = []
for (nL, nM) in ((6, 3), (6, 5), (12, 5), (20, 10), (40, 12)):
    F = 256
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM, seed=0xC0FFEE + nL * 100 + nM)
    nR = nL + nM
    have_ref = os.path.exists(ol.REF_ASSIGN_OFAST_SO)
    # the reference on one core
    t0 = time.perf_counter()
    for f in frames[:128]:
        if have_ref:
            c, ridx = ol.ref_condition_costs(f, nR, nM)
            ol.ref_assignment_prob(c, len(ridx) - nM, nM, k, ofast=True)
        else:
            c, ridx = ol.condition_costs(f, nR, nM)
            ol.assignment_prob(c, len(ridx) - nM, nM, k)
    cpu_us = 1e6 * (time.perf_counter() - t0) / 128
    line = {"nL": nL, "nM": nM, "cpu_us_per_frame": cpu_us, "cpu_kind": "reference" if have_ref else "port", "gpu": {}}
    cross = None
    for B in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        raw = np.ascontiguousarray(np.concatenate(frames[:B]))
        h_nL = np.full(B, nL, np.int32); h_nM = np.full(B, nM, np.int32)
        coff = np.arange(B, dtype=np.int64) * nR * nM; poff = np.arange(B, dtype=np.int64) * nM * (nL + 1)
        probs = np.zeros(B * nM * (nL + 1)); nf = np.zeros(B, np.int32)
        n = 300 if B == 1 else 60
        for i in range(-(300 if B == 1 else 20), n):
            if i == 0: t0 = time.perf_counter()
            rc = lib.kbest_assoc_probs_batch_f64(ctx, B, p(h_nL), p(h_nM), p(raw), p(coff), k, p(probs), p(poff), p(nf))
        per_call = 1e6 * (time.perf_counter() - t0) / n
        assert rc == 0
        line["gpu"][B] = {"us_per_call": per_call, "us_per_frame": per_call / B}
        if cross is None and per_call / B < cpu_us:
            cross = B
    line["gpu_overtakes_one_core_at_B"] = cross
    rows.append(line)
    print(f"nL={nL:2d} nM={nM:2d}  CPU {cpu_us:7.1f} us/frame | " + " ".join(f"B={B}: {v['us_per_frame']:.1f}" for B, v in line["gpu"].items()) + f" | crossover B={cross}", flush=True)
if len(sys.argv) > 1:
    json.dump({"k": k, "rows": rows, "what": "us per frame: reference getAssignmentProbs chain on one host core vs kbest_assoc_probs_batch_f64 with B frames per call"}, open(sys.argv[1], "w"), indent=1)
