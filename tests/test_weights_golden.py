"""Weights functions against golden vectors recorded from the reference's OWN code
(tests/golden/weights_golden.npz, made by tests/golden/gen_weights_golden.py from verbatim slices of
assignment.cpp compiled with -O2): conditionCosts (:439-525), toProbs (:527-542), assignmentProb (:547-683),
bruteForceProb (:835-964).

CPU part: the oracle restatement must reproduce them bit for bit (same libm exp, same summation order).
GPU part: the engine through the C ABI -- conditionCosts bit-exact, probabilities within 1e-9 RELATIVE (the device
exp differs from libm by <= 1 ulp, measured agreement ~1e-15; north-star tolerance 1e-6 relative), including the fused
one-launch association path of kbest_assoc_probs_batch_f64 on the RAW blocks.
"""
import os

import numpy as np
import pytest

import oracle_lib as ol

RTOL, ATOL = 1e-9, 1e-300  # relative, as the north star states its tolerance (an absolute one says nothing about 1e-15 weights)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "weights_golden.npz")


def _cases():
    z = np.load(GOLD)
    for name in z["names"]:
        name = str(name)
        nL, nM, k, good, brute = (int(x) for x in z[name + "/meta"])
        yield dict(name=name, nL=nL, nM=nM, k=k, good=good, raw=z[name + "/raw"], cond=z[name + "/cond"],
                   rowIdx=z[name + "/rowIdx"].astype(np.int64), probs=z[name + "/probs"], toProbs=z[name + "/toProbs"],
                   brute=z[name + "/brute"] if brute else None)


CASES = list(_cases())


def _scatter_back(c):
    """getAssignmentProbs' scatter into the original landmark numbering (assignment.cpp:68-74)."""
    nL, nM, condL = c["nL"], c["nM"], c["good"] - c["nM"]
    p = np.zeros((nM, nL + 1))
    for m in range(nM):
        for l in range(condL):
            p[m, c["rowIdx"][l]] = c["probs"][m, l]
        p[m, nL] = c["probs"][m, condL]
    return p


def test_oracle_condition_costs_matches_reference_golden():
    for c in CASES:
        cond, idx = ol.condition_costs(c["raw"], c["nL"] + c["nM"], c["nM"])
        assert idx.tolist() == c["rowIdx"].tolist(), c["name"]
        assert cond.view(np.int64).tolist() == c["cond"].view(np.int64).tolist(), c["name"]


def test_oracle_assignment_prob_matches_reference_golden():
    for c in CASES:
        condL = c["good"] - c["nM"]
        p, nf = ol.assignment_prob(c["cond"], condL, c["nM"], c["k"])
        assert p.shape == c["probs"].shape, c["name"]
        assert p.view(np.int64).tolist() == c["probs"].view(np.int64).tolist(), c["name"]


def test_oracle_brute_force_and_to_probs_match_reference_golden():
    for c in CASES:
        tp = c["cond"].copy()
        ol.oracle().orc_to_probs(tp, tp.size)
        assert tp.view(np.int64).tolist() == c["toProbs"].view(np.int64).tolist(), c["name"]
        if c["brute"] is not None:
            pb, nf, uk = ol.brute_force_prob(c["cond"], c["good"] - c["nM"], c["nM"])
            assert pb.view(np.int64).tolist() == c["brute"].view(np.int64).tolist(), c["name"]


@pytest.mark.skipif(not ol.have_ref_assign(), reason="oracle/_ref/libref_assign.so not built (needs /root/reference)")
def test_golden_is_what_the_compiled_reference_returns():
    for c in CASES:
        cond, idx = ol.ref_condition_costs(c["raw"], c["nL"] + c["nM"], c["nM"])
        assert idx.tolist() == c["rowIdx"].tolist() and cond.tolist() == c["cond"].tolist(), c["name"]
        p = ol.ref_assignment_prob(cond, len(idx) - c["nM"], c["nM"], c["k"])
        assert p.view(np.int64).tolist() == c["probs"].view(np.int64).tolist(), c["name"]


@pytest.mark.skipif(not ol.have_ref_assign(), reason="oracle/_ref/libref_assign.so not built (needs /root/reference)")
def test_oracle_weights_vs_compiled_reference_random_frames():
    from probabilisticsemslam_amd import workloads as wl
    for seed, nL, nM, k in ((1, 20, 10, 200), (2, 8, 4, 60), (3, 30, 6, 150), (4, 5, 5, 500), (5, 14, 1, 10)):
        for f in wl.kitti_like_frames(12, nL=nL, nM=nM, seed=0xABC000 + seed):
            cond, idx = ol.condition_costs(f, nL + nM, nM)
            rc, ri = ol.ref_condition_costs(f, nL + nM, nM)
            assert idx.tolist() == ri.tolist() and cond.tolist() == rc.tolist()
            condL = len(idx) - nM
            p, _ = ol.assignment_prob(cond, condL, nM, k)
            pr = ol.ref_assignment_prob(cond, condL, nM, k)
            assert p.view(np.int64).tolist() == pr.view(np.int64).tolist()


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_condition_costs_matches_reference_golden(engine):
    conds, idxs = engine.condition_costs([c["raw"] for c in CASES], [c["nL"] + c["nM"] for c in CASES],
                                         [c["nM"] for c in CASES])
    for c, cond, idx in zip(CASES, conds, idxs):
        assert idx.tolist() == c["rowIdx"].tolist(), c["name"]
        assert cond.view(np.int64).tolist() == c["cond"].view(np.int64).tolist(), c["name"]


@pytest.mark.gpu
def test_gpu_assignment_prob_matches_reference_golden(engine):
    for c in CASES:
        condL = c["good"] - c["nM"]
        out, nf = engine.weights([c["cond"]], [condL], [c["nM"]], c["k"])
        want = c["probs"][:, : condL + 1]  # nM == 1: the reference's row is 1 x size(cost), zeros beyond nL+1
        np.testing.assert_allclose(out[0], want, rtol=RTOL, atol=ATOL, err_msg=c["name"])
        if c["brute"] is not None:
            outb, _ = engine.weights([c["cond"]], [condL], [c["nM"]], 20000, brute_force=True)
            np.testing.assert_allclose(outb[0], c["brute"][:, : condL + 1], rtol=RTOL, atol=ATOL, err_msg=c["name"])


@pytest.mark.gpu
def test_gpu_assoc_probs_on_raw_blocks_matches_reference_golden(engine):
    # one call per distinct k (k is a launch parameter), frames of different shapes in the same batch
    for k in sorted({c["k"] for c in CASES}):
        sel = [c for c in CASES if c["k"] == k]
        out, nf = engine.weights([c["raw"] for c in sel], [c["nL"] for c in sel], [c["nM"] for c in sel], k,
                                 condition=True)
        for c, p in zip(sel, out):
            np.testing.assert_allclose(p, _scatter_back(c), rtol=RTOL, atol=ATOL, err_msg=c["name"])
    # and one frame per call, the reference's own call pattern (system.cpp:268)
    for c in CASES[:8]:
        out, nf = engine.weights([c["raw"]], [c["nL"]], [c["nM"]], c["k"], condition=True)
        np.testing.assert_allclose(out[0], _scatter_back(c), rtol=RTOL, atol=ATOL, err_msg=c["name"])
