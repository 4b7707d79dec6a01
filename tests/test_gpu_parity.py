"""GPU: the HIP engine, called through the C ABI, against (a) the golden vectors recorded from the
unmodified reference solver, (b) the CPU oracle on seeded inputs, (c) size-independent properties at
BASELINE.json's full sizes.  Bit-exact for nf / row4col / col4row / gains; weights within 1e-12
(north star: 1e-6 relative)."""
import numpy as np
import pytest

import oracle_lib as ol
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.fixture(scope="module")
def eng():
    return pk.KBestEngine(0)


def check_against(eng_out, want_nf, want_r4c, want_c4r, want_g, tag=""):
    nf, r4c, c4r, g = eng_out
    assert nf == want_nf, (tag, nf, want_nf)
    assert (r4c[:nf] == want_r4c[:nf]).all(), tag
    # which zero-padded column a left-over row sits on is an artefact of exact-tie resolution
    # among identical columns -- col4row is compared after mapping every value >= numCol to -1 ...
    M = r4c.shape[-1]
    assert (ol.canon_col4row(c4r[:nf], M) == ol.canon_col4row(want_c4r[:nf], M)).all(), tag
    # ... and must still be a complete assignment of the padded square problem: a permutation of 0..numRow-1
    if nf:
        assert (np.sort(np.asarray(c4r[:nf]), axis=-1) == np.arange(c4r.shape[-1])).all(), tag
    assert (bits(g[:nf]) == bits(want_g[:nf])).all(), tag


def test_golden_vectors(eng, golden):
    for name in golden.names:
        c = golden.case(name)
        nf, r4c, c4r, g = eng.kbest(c["cost"].reshape(1, -1), c["N"], c["M"], c["k"], c["maximize"], c["cutoff"])
        check_against((int(nf[0]), r4c[0], c4r[0], g[0]), c["nf"], c["row4col"], c["col4row"], c["gain"], name)


def test_golden_vectors_no_prune(eng, golden):
    # early termination off: same answers, and the push count equals the oracle's (SURVEY 8(d) "P")
    for name in golden.names:
        c = golden.case(name)
        nf, r4c, c4r, g, pushed = eng.kbest(c["cost"].reshape(1, -1), c["N"], c["M"], c["k"], c["maximize"],
                                            c["cutoff"], count_pushed=True, prune=False)
        check_against((int(nf[0]), r4c[0], c4r[0], g[0]), c["nf"], c["row4col"], c["col4row"], c["gain"], name)
        st = ol.orc_kbest(c["cost"], c["N"], c["M"], c["k"], c["maximize"], c["cutoff"], want_stats=True)[4]
        assert int(pushed[0]) == st.children_pushed, name


def test_golden_as_one_ragged_batch(eng, golden):
    # all golden cases with the same (k, maximize, cutoff) in ONE launch with per-problem shapes
    groups = {}
    for name in golden.names:
        c = golden.case(name)
        groups.setdefault((c["k"], c["maximize"], c["cutoff"]), []).append(c)
    for (k, maximize, cutoff), cs in groups.items():
        nRow = np.array([c["N"] for c in cs], np.int32)
        nCol = np.array([c["M"] for c in cs], np.int32)
        sizes = nRow.astype(np.int64) * nCol
        off = np.zeros(len(cs), np.int64)
        off[1:] = np.cumsum(sizes)[:-1]
        flat = np.concatenate([c["cost"] for c in cs])
        N, M = int(nRow.max()), int(nCol.max())
        nf, r4c, c4r, g = eng.kbest(flat, N, M, k, maximize, cutoff, nRow=nRow, nCol=nCol, costOff=off)
        for i, c in enumerate(cs):
            check_against((int(nf[i]), r4c[i][:, :c["M"]], c4r[i][:, :c["N"]], g[i]), c["nf"], c["row4col"],
                          c["col4row"], c["gain"], c["name"])


@pytest.mark.parametrize("name,nb", [("c1", 1), ("c2", 64), ("c3", 24), ("c4", 8)])
def test_dense_configs_vs_oracle(eng, name, nb):
    costs, N, M, k = wl.dense_config(name, B=nb)
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (nf == onf).all()
    assert (r4c == or4c).all() and (c4r == oc4r).all()
    assert (bits(g) == bits(og)).all()


def test_random_shapes_vs_oracle(eng):
    rng = np.random.default_rng(2024)
    for trial in range(60):
        N = int(rng.integers(1, 65))
        M = int(rng.integers(1, N + 1))
        k = int(rng.integers(1, 80))
        B = int(rng.integers(1, 6))
        costs = rng.random((B, N * M)) * 20 - 5
        mode = trial % 4
        if mode == 1:
            costs[rng.random((B, N * M)) < 0.4] = np.inf
        maximize = mode == 2
        cutoff = [None, None, None, 4.0][mode]
        nf, r4c, c4r, g = eng.kbest(costs, N, M, k, maximize, cutoff)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, maximize, cutoff)
        assert (nf == onf).all(), trial
        for b in range(B):
            n = nf[b]
            check_against((n, r4c[b], c4r[b], g[b]), onf[b], or4c[b], oc4r[b], og[b], trial)


def test_ties_multiset(eng):
    # integer costs: equal-gain hypotheses have unspecified relative order:
    # the multiset of gains and the validity / distinctness of assignments must still hold
    rng = np.random.default_rng(5)
    for trial in range(20):
        N = int(rng.integers(2, 10)); M = int(rng.integers(1, N + 1)); k = int(rng.integers(1, 50))
        cost = rng.integers(0, 4, N * M).astype(np.float64)
        nf, r4c, c4r, g = eng.kbest(cost.reshape(1, -1), N, M, k)
        onf, or4c, oc4r, og = ol.orc_kbest(cost, N, M, k)
        n = int(nf[0])
        assert n == onf
        assert sorted(g[0, :n].tolist()) == sorted(og[:n].tolist())
        seen = set()
        for s in range(n):
            rows = tuple(r4c[0, s].tolist())
            assert len(set(rows)) == M and rows not in seen
            seen.add(rows)
            assert g[0, s] == sum(cost[c * N + r] for c, r in enumerate(rows))


@pytest.mark.parametrize("name", ["c2", "c3", "c4"])
def test_full_size_properties(eng, name):
    """BASELINE.json full sizes: gains non-decreasing, assignments are distinct injections, each gain is the
    serial column-order sum of the chosen entries (bit-exact), nf == k, and the whole batch equal to the oracle's."""
    costs, N, M, k = wl.dense_config(name)
    B = costs.shape[0]
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
    assert (nf == k).all()
    assert (np.diff(g, axis=1) >= 0).all()
    # rows distinct per solution
    srt = np.sort(r4c, axis=2)
    assert (np.diff(srt, axis=2) > 0).all()
    # col4row is the inverse of row4col
    bi, si, ci = np.meshgrid(np.arange(B), np.arange(k), np.arange(M), indexing="ij")
    assert (c4r[bi, si, r4c] == ci).all()
    # gain = serial sum in column order of the shifted costs + CDelta*M (kBest2D cpp:583-600)
    cd = costs.min(axis=1)
    shifted = costs - cd[:, None]
    acc = np.zeros((B, k))
    for c in range(M):
        acc = acc + shifted[np.arange(B)[:, None], c * N + r4c[:, :, c]]
    assert (bits(acc + (cd * M)[:, None]) == bits(g)).all()
    # solutions of one problem are pairwise distinct
    for b in range(0, B, max(1, B // 64)):
        assert len({tuple(x) for x in r4c[b].tolist()}) == k
    # against the checker: EVERY matrix of every config (c3: 4 096 matrices, ~20 s of host time; c4: 1 024, ~20 s)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (nf == onf).all() and (r4c == or4c).all() and (bits(g) == bits(og)).all()


def test_weights_kitti_like(eng):
    frames = wl.kitti_like_frames(40)
    conds, nLs, nMs = [], [], []
    for f in frames:
        cond, idx = ol.condition_costs(f, 30, 10)
        conds.append(cond); nLs.append(len(idx) - 10); nMs.append(10)
    probs, nf = eng.weights(conds, nLs, nMs, 200)
    for i, cond in enumerate(conds):
        po, onf = ol.assignment_prob(cond, nLs[i], 10, 200)
        assert nf[i] == onf
        np.testing.assert_allclose(probs[i], po, rtol=1e-12, atol=1e-15)   # north star: 1e-6 relative
    # SURVEY 8(c) weight KAT
    assert abs(probs[0][0][9] - 0.61485235124407667) < 1e-14


def test_weights_small_vs_permanent(eng):
    # config 5's check: weights equal the exact permanent ratio on small exhaustive sub-problems
    frames = wl.kitti_like_frames(12, nL=6, nM=3)
    conds, nLs = [], []
    for f in frames:
        cond, idx = ol.condition_costs(f, 9, 3)
        conds.append(cond); nLs.append(len(idx) - 3)
    probs, nf = eng.weights(conds, nLs, [3] * len(conds), 200)
    for i, cond in enumerate(conds):
        nR = nLs[i] + 3
        A = np.exp(-cond.reshape(3, nR).T)
        Z = ol.permanent(A)
        want = np.zeros((3, nLs[i] + 1))
        for c in range(3):
            for r in range(nR):
                if A[r, c] > 0:
                    want[c, min(r, nLs[i])] += A[r, c] * ol.permanent(np.delete(np.delete(A, r, 0), c, 1)) / Z
        np.testing.assert_allclose(probs[i], want, rtol=0, atol=1e-9)


def test_weights_single_column(eng):
    cost = np.array([0.5, 43.0, 2.0, 10.0])
    p = pk.assignmentProb(cost, 3, 1, 200)
    po, _ = ol.assignment_prob(cost, 3, 1, 200)
    np.testing.assert_allclose(p, po, rtol=1e-14)


def test_reference_named_mirrors(eng):
    cs, N, M, k = wl.dense_config("c1")
    nf, c4r, r4c, g = pk.kBest2D(k, N, M, False, cs[0])
    assert nf == 10 and r4c[0].tolist() == [0, 2, 7, 5, 3, 4, 6, 1] and float.hex(g[0]) == "0x1.0bd6f90d82018p+0"
    nf2, _, r4c2, g2 = pk.kBest2DCutoff(k, N, M, False, cs[0], 0.1)
    onf, or4c, _, og = ol.orc_kbest(cs[0], N, M, k, cutoff=0.1)
    assert nf2 == onf and (r4c2[:nf2] == or4c[:nf2]).all()


def test_unsupported_and_bad_args(eng):
    with pytest.raises(pk.KBestError):
        eng.kbest(np.zeros((1, 16385 * 2)), 16385, 2, 2)    # beyond KBEST_MAX_DIM_EXACT: loud, no fallback
    with pytest.raises(pk.KBestError):
        eng.kbest(np.zeros((1, 6)), 2, 3, 2)                # numRow < numCol


def test_subtree_sharding_merges_to_global_kbest(eng):
    """Multi-GPU latency mode: rank g expands only root children on columns c % G == g; the k
    smallest of {root} U per-rank lists equal the single-GPU k-best."""
    costs, N, M, k = wl.dense_config("c3", B=3)
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
    for G in (2, 4):
        parts = [eng.kbest(costs, N, M, k, root_shard=(r, G)) for r in range(G)]
        for b in range(costs.shape[0]):
            cand = [(g[b, 0], tuple(r4c[b, 0]))]
            for pnf, pr4c, pc4r, pg in parts:
                cand += [(pg[b, s], tuple(pr4c[b, s])) for s in range(1, pnf[b])]
            cand.sort(key=lambda t: t[0])
            assert [c[0] for c in cand[:k]] == g[b].tolist()
            assert [c[1] for c in cand[:k]] == [tuple(x) for x in r4c[b]]


def test_condition_costs_device(eng):
    # conditionCosts on the device vs the oracle restatement (assignment.cpp:439-525): bit-exact
    frames = wl.kitti_like_frames(30) + wl.kitti_like_frames(10, nL=6, nM=3)
    shapes = [(30, 10)] * 30 + [(9, 3)] * 10
    conds, ridx = eng.condition_costs(frames, [s[0] for s in shapes], [s[1] for s in shapes])
    for f, (nR, nC), c, r in zip(frames, shapes, conds, ridx):
        oc, orx = ol.condition_costs(f, nR, nC)
        assert (r == orx).all()
        assert (bits(c) == bits(oc)).all()


def test_assoc_probs_pipeline(eng):
    """cost block in -> probabilities out on the device (conditionCosts -> assignmentProb -> scatter back),
    against the same chain of the oracle (getAssignmentProbs, assignment.cpp:57-74)."""
    frames = wl.kitti_like_frames(60)
    nL, nM = 20, 10
    probs, nf = eng.weights(frames, [nL] * len(frames), [nM] * len(frames), 200, condition=True)
    for i, f in enumerate(frames):
        cond, idx = ol.condition_costs(f, nL + nM, nM)
        condL = len(idx) - nM
        pc, onf = ol.assignment_prob(cond, condL, nM, 200)
        want = np.zeros((nM, nL + 1))
        want[:, idx[:condL]] = pc[:, :condL]
        want[:, nL] = pc[:, condL]
        assert nf[i] == onf
        np.testing.assert_allclose(probs[i], want, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(probs[i].sum(axis=1), 1.0, rtol=1e-12)


def test_config5_streamed_frames(eng):
    """BASELINE config 5: a stream of per-frame 30-row x 10-column (-> 30x30 padded) cost matrices, k = 200,
    cutoff 42 active, processed chunk by chunk as a frame loop would hand them over.  Every 8th frame is checked
    against the oracle chain; every frame's probabilities must be a distribution per measurement."""
    frames = wl.kitti_like_frames(1000)
    nL, nM, chunk = 20, 10, 125
    for c0 in range(0, len(frames), chunk):
        fs = frames[c0:c0 + chunk]
        probs, nf = eng.weights(fs, [nL] * len(fs), [nM] * len(fs), 200, condition=True)
        for i, f in enumerate(fs):
            assert nf[i] >= 1
            np.testing.assert_allclose(probs[i].sum(axis=1), 1.0, rtol=1e-12)
            assert (probs[i] >= 0).all()
            if (c0 + i) % 8 == 0:
                cond, idx = ol.condition_costs(f, nL + nM, nM)
                condL = len(idx) - nM
                pc, onf = ol.assignment_prob(cond, condL, nM, 200)
                want = np.zeros((nM, nL + 1))
                want[:, idx[:condL]] = pc[:, :condL]
                want[:, nL] = pc[:, condL]
                assert nf[i] == onf
                np.testing.assert_allclose(probs[i], want, rtol=1e-12, atol=1e-15)


def test_cpp_shims_drop_in(eng, tmp_path):
    """A C++ caller written like the reference's own call sites, compiled against include/kbest_shims.hpp and
    linked to libkbest_amd.so (the link-level drop-in of INTEGRATION.md), gives the checker's answers."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shim_drop_in")
    libdir = os.path.join(root, "probabilisticsemslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "shim_drop_in.cpp"), "-o", exe,
                           "-L", libdir, "-l:libkbest_amd.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-L/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.check_output([exe, "8", "8", "10", "12345"], text=True).splitlines()
    cs, N, M, k = wl.dense_config("c1")
    onf, or4c, oc4r, og = ol.orc_kbest(cs[0], N, M, k)
    assert out[0] == f"kBest2D nf {onf}"
    for s in range(onf):
        want = "g " + float.hex(og[s]) + " r4c " + " ".join(map(str, or4c[s])) + " c4r " + " ".join(map(str, oc4r[s]))
        got = out[1 + s].split()
        assert float.fromhex(got[1]) == og[s] and " ".join(got[2:]) == " ".join(want.split()[2:])
    cnf, cr4c, _, cg = ol.orc_kbest(cs[0], N, M, k, cutoff=0.1)
    i = 1 + onf
    assert out[i] == f"kBest2DCutoff nf {cnf} toCut 1"
    assert [float.fromhex(l.split()[1]) for l in out[i + 1: i + 1 + cnf]] == cg[:cnf].tolist()
    i += 1 + cnf
    a = out[i].split()
    assert a[:3] == ["assign2D", "ok", "1"] and float.fromhex(a[4]) == og[0] and list(map(int, a[6:])) == or4c[0].tolist()
    frame = wl.kitti_like_frames(1, nL=6, nM=3)[0]
    cond, idx = ol.condition_costs(frame, 9, 3)
    assert out[i + 1] == f"conditionCosts rows {len(idx)} idx " + " ".join(map(str, idx))
    po, _ = ol.assignment_prob(cond, len(idx) - 3, 3, 200)
    for m in range(3):
        got = np.array([float.fromhex(x) for x in out[i + 2 + m].split()[1:]])
        np.testing.assert_allclose(got, po[m], rtol=1e-12, atol=1e-15)
    qo, _, _ = ol.brute_force_prob(cond, len(idx) - 3, 3)
    for m in range(3):
        assert out[i + 5 + m].startswith("q")
        got = np.array([float.fromhex(x) for x in out[i + 5 + m].split()[1:]])
        np.testing.assert_allclose(got, qo[m], rtol=1e-12, atol=1e-15)
    # toProbs, assign2D (rectangular / maximise / infeasible, with duals), shortestPathCPP: against the goldens
    # recorded from the reference itself
    zw = np.load(os.path.join(root, "tests", "golden", "weights_golden.npz"))
    za = np.load(os.path.join(root, "tests", "golden", "assign_golden.npz"))
    i += 8
    assert out[i].startswith("t ")
    tp = cond.copy()
    ol.oracle().orc_to_probs(tp, tp.size)
    np.testing.assert_allclose([float.fromhex(x) for x in out[i].split()[1:]], tp, rtol=1e-14, atol=0)
    assert zw["small_f0/toProbs"].tolist() == tp.tolist()  # (the block is golden frame small_f0)

    def fields(line, keys):
        toks = line.split()
        pos = {k: toks.index(k) for k in keys}
        order = sorted(pos.values()) + [len(toks)]
        return {k: toks[pos[k] + 1: order[order.index(pos[k]) + 1]] for k in keys}

    f = fields(out[i + 1], ["ok", "g", "solved", "activeCol", "r4c", "c4r", "u", "v", "forb"])
    assert out[i + 1].startswith("assign2D_rect") and f["ok"] == ["1"] and f["solved"] == ["1"] and f["activeCol"] == ["0"]
    assert float.fromhex(f["g"][0]) == float(za["rect_12x5_max/gain"][0])
    assert list(map(int, f["r4c"])) == za["rect_12x5_max/row4col"].tolist()
    assert list(map(int, f["c4r"])) == za["rect_12x5_max/col4row"].tolist()
    assert [float.fromhex(x) for x in f["u"]] == za["rect_12x5_max/u"].tolist()
    assert [float.fromhex(x) for x in f["v"]] == za["rect_12x5_max/v"].tolist()
    forb = [0] * 12
    forb[int(za["rect_12x5_max/row4col"][0])] = 1
    assert list(map(int, f["forb"])) == forb                     # cpp:235
    assert out[i + 2] == "assign2D_infeasible ok 0"
    f = fields(out[i + 3], ["rc", "g", "r4c", "u"])
    assert out[i + 3].startswith("shortestPathCPP ") and f["rc"] == ["0"]
    assert float.fromhex(f["g"][0]) == float(za["spc_10x10_g4/gain"][0])
    assert list(map(int, f["r4c"])) == za["spc_10x10_g4/row4col"].tolist()
    assert [float.fromhex(x) for x in f["u"]] == za["spc_10x10_g4/u"].tolist()
    f = fields(out[i + 4], ["rc", "g"])
    assert f["rc"] == ["1"] and float.fromhex(f["g"][0]) == -1.0   # cpp:197-203


def test_quadric_costs_and_full_association_chain(eng):
    """Rows f2 + f1: (mean, covariance) pairs -> Mahalanobis cost blocks -> conditionCosts -> assignmentProb ->
    scatter back, all on the device, against the same chain of the checker."""
    from test_cost_builders import synth_quadric_frame
    rng = np.random.default_rng(42)
    frames = [synth_quadric_frame(rng, int(rng.integers(1, 30)), int(rng.integers(1, 11))) for _ in range(40)]
    costs = eng.quadric_costs(frames, 10.0)
    for f, c in zip(frames, costs):
        oc = ol.quadric_costs(*f, 10.0)
        assert (np.isinf(c) == np.isinf(oc)).all()
        fin = np.isfinite(oc)
        np.testing.assert_allclose(c[fin], oc[fin], rtol=1e-13, atol=0)      # same operation order: ~bit-equal
    probs, nf = eng.quadric_assoc_probs(frames, 10.0, 200)
    for i, f in enumerate(frames):
        nL, nM = len(f[0]), len(f[2])
        oc = ol.quadric_costs(*f, 10.0)
        cond, idx = ol.condition_costs(oc, nL + nM, nM)
        condL = len(idx) - nM
        pc, onf = ol.assignment_prob(cond, condL, nM, 200)
        want = np.zeros((nM, nL + 1))
        want[:, idx[:condL]] = pc[:, :condL]
        want[:, nL] = pc[:, condL]
        np.testing.assert_allclose(probs[i], want, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(probs[i].sum(axis=1), 1.0, rtol=1e-12)


def test_bounding_box_matching(eng):
    """Row f4: asgnBB (IoU profits, -inf fill, gate profit, kBest2D k=1 maximize) on the device vs the checker."""
    from test_cost_builders import synth_boxes
    rng = np.random.default_rng(7)
    L, R = [], []
    for _ in range(60):
        nL, nR = int(rng.integers(1, 20)), int(rng.integers(0, 20))
        l, r = synth_boxes(rng, nL, nR)
        L.append(l); R.append(r)
    got = eng.bb_match(L, R, 0.2)
    for l, r, g in zip(L, R, got):
        assert g.tolist() == ol.asgn_bb(l, r, 0.2).tolist()


def test_comp_methods_harness(eng, tmp_path):
    """Row f3: the reference's compMethods protocol (file format round trip, k sweep, error vs brute force) with
    its own acceptance rule: no frame above 0.1 (comparison.cpp:319), exhaustive small frames exact at k >= 100."""
    import harness_comp_methods as h
    table = h.run(n_frames=48, nL=6, nM=3, directory=str(tmp_path), verbose=False)
    # the hard limit applies to the configured k (200; comparison.cpp:319-324); k = 1 and 20 are only plotted
    assert (table[200]["err"] <= 0.1).all() and (table[1000]["err"] <= 0.1).all()
    assert (table[200]["err"] <= 1e-9).all() and (table[1000]["err"] <= 1e-9).all()   # exhaustive here: exact
    worst = [table[k]["err"].max() for k in (1, 20, 100, 200)]
    assert worst == sorted(worst, reverse=True)                  # more assignments never hurt


def test_large_maps_condition_down_to_solver_size(eng):
    """getAssignmentProbs sees EVERY landmark of the map: the raw (nL+nM) x nM matrix can have hundreds of rows.
    conditionCosts runs on the device for any row count; what it keeps goes to the LDS kernel (<= 64 rows) or to
    the general-size kernel (<= 1 024 rows) of the same launch."""
    from test_cost_builders import synth_quadric_frame
    rng = np.random.default_rng(99)
    frames = []
    for _ in range(12):
        nL, nM = int(rng.integers(150, 400)), int(rng.integers(2, 8))
        lm, lc, mm, mc = synth_quadric_frame(rng, nL, nM)
        frames.append((lm * 8.0, lc, mm * 8.0, mc))          # spread out: most landmarks are far from every measurement
    probs, nf = eng.quadric_assoc_probs(frames, 10.0, 200)
    checked = 0
    for i, f in enumerate(frames):
        nL, nM = len(f[0]), len(f[2])
        oc = ol.quadric_costs(*f, 10.0)
        cond, idx = ol.condition_costs(oc, nL + nM, nM)
        if len(idx) > 1024:
            assert nf[i] == -1 and (probs[i] == 0).all()      # loud, not wrong
            continue
        condL = len(idx) - nM
        pc, onf = ol.assignment_prob(cond, condL, nM, 200)
        want = np.zeros((nM, nL + 1))
        want[:, idx[:condL]] = pc[:, :condL]
        want[:, nL] = pc[:, condL]
        assert nf[i] == onf
        np.testing.assert_allclose(probs[i], want, rtol=1e-9, atol=1e-12)
        checked += 1
    assert checked >= 8
    # and conditionCosts alone on a tall matrix, bit-exact
    f = frames[0]
    oc = ol.quadric_costs(*f, 10.0)
    nR, nC = len(f[0]) + len(f[2]), len(f[2])
    conds, ridx = eng.condition_costs([oc], [nR], [nC])
    wc, wi = ol.condition_costs(oc, nR, nC)
    assert (ridx[0] == wi).all() and (bits(conds[0]) == bits(wc)).all()


def test_large_k_brute_force_style(eng):
    """bruteForceProb drives kBest2D with k in the thousands on small matrices (assignment.cpp:858-880): the pool
    then exceeds one entry per thread and the launch shape has to grow with k."""
    rng = np.random.default_rng(17)
    for N, M, k in ((9, 4, 1500), (12, 5, 3000), (7, 7, 4000), (30, 6, 2500)):
        costs = rng.random((3, N * M)) * 8
        nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
        assert (nf == onf).all(), (N, M, k)
        for b in range(3):
            n = nf[b]
            assert (r4c[b, :n] == or4c[b, :n]).all() and (bits(g[b, :n]) == bits(og[b, :n])).all(), (N, M, k)


def _same_as_oracle(eng, costs, N, M, k, maximize=False, cutoff=None, tag=None, **kw):
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k, maximize, cutoff, **kw)[:4]
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(np.asarray(costs).reshape(len(nf), -1), N, M, k, maximize, cutoff)
    assert (nf == onf).all(), (tag, nf, onf)
    for b in range(len(nf)):
        n = nf[b]
        check_against((n, r4c[b], c4r[b], g[b]), onf[b], or4c[b], oc4r[b], og[b], tag)


def test_general_size_kernel_on_small_problems(eng, golden, monkeypatch):
    """kbest_wide.hip (hypotheses and pool in HBM, R rows per lane) forced onto problems the LDS kernel normally
    takes: golden vectors of the reference and random shapes, bit-exact, and the reference's push count."""
    monkeypatch.setenv("KBEST_FORCE_WIDE", "1")
    for name in golden.names:
        c = golden.case(name)
        nf, r4c, c4r, g, pushed = eng.kbest(c["cost"].reshape(1, -1), c["N"], c["M"], c["k"], c["maximize"],
                                            c["cutoff"], count_pushed=True, prune=False)
        check_against((int(nf[0]), r4c[0], c4r[0], g[0]), c["nf"], c["row4col"], c["col4row"], c["gain"], name)
        st = ol.orc_kbest(c["cost"], c["N"], c["M"], c["k"], c["maximize"], c["cutoff"], want_stats=True)[4]
        assert int(pushed[0]) == st.children_pushed, name
        nf, r4c, c4r, g = eng.kbest(c["cost"].reshape(1, -1), c["N"], c["M"], c["k"], c["maximize"], c["cutoff"])
        check_against((int(nf[0]), r4c[0], c4r[0], g[0]), c["nf"], c["row4col"], c["col4row"], c["gain"], name)
    rng = np.random.default_rng(77)
    for trial in range(24):
        N = int(rng.integers(1, 65)); M = int(rng.integers(1, N + 1)); k = int(rng.integers(1, 60)); B = int(rng.integers(1, 5))
        costs = rng.random((B, N * M)) * 20 - 5
        mode = trial % 4
        if mode == 1:
            costs[rng.random((B, N * M)) < 0.4] = np.inf
        _same_as_oracle(eng, costs, N, M, k, mode == 2, [None, None, None, 4.0][mode], tag=trial)


def test_more_than_64_rows(eng):
    """numRow beyond the LDS kernel (the reference has no size limit): 2, 4 and 8 rows per lane."""
    rng = np.random.default_rng(4242)
    for N, M, k, B in ((65, 65, 12, 2), (100, 37, 40, 3), (128, 128, 25, 2), (130, 5, 200, 2), (200, 90, 15, 2),
                       (257, 20, 30, 1), (300, 300, 4, 1), (512, 16, 20, 1)):
        costs = rng.random((B, N * M)) * 30
        if N == 100:
            costs[rng.random((B, N * M)) < 0.5] = np.inf
        _same_as_oracle(eng, costs, N, M, k, maximize=(N == 200), cutoff=(2.0 if N == 130 else None), tag=(N, M, k))


def test_mixed_batch_small_and_large_rows(eng):
    """one ragged launch with problems on both sides of 64 rows: the LDS kernel and the general-size kernel each
    take their share and write into the same output tables."""
    rng = np.random.default_rng(31)
    shapes = [(10, 4), (70, 9), (64, 64), (96, 50), (3, 3), (65, 2), (40, 11), (150, 30)]
    k = 30
    nRow = np.array([s[0] for s in shapes], np.int32); nCol = np.array([s[1] for s in shapes], np.int32)
    blocks = [rng.random(n * m) * 10 for n, m in shapes]
    off = np.zeros(len(shapes), np.int64); off[1:] = np.cumsum([len(x) for x in blocks])[:-1]
    N, M = int(nRow.max()), int(nCol.max())
    nf, r4c, c4r, g = eng.kbest(np.concatenate(blocks), N, M, k, nRow=nRow, nCol=nCol, costOff=off)
    for i, (n, m) in enumerate(shapes):
        onf, or4c, oc4r, og = ol.orc_kbest(blocks[i], n, m, k)
        check_against((int(nf[i]), r4c[i][:, :m], c4r[i][:, :n], g[i]), onf, or4c, oc4r, og, (n, m))


def test_k_beyond_the_lds_pool(eng):
    """bruteForceProb asks kBest2D for up to 20 000 assignments (assignment.cpp:868): the pool then lives in HBM."""
    rng = np.random.default_rng(8)
    for N, M, k in ((8, 8, 20000), (10, 6, 20000), (64, 64, 6000)):
        costs = rng.random((1, N * M)) * 6
        _same_as_oracle(eng, costs, N, M, k, tag=(N, M, k))
    # every assignment of a 7 x 7 problem (5040 of them): the enumeration ends by itself
    costs = rng.random((1, 49)) * 3
    nf, r4c, c4r, g = eng.kbest(costs, 7, 7, 20000)
    assert nf[0] == 5040 and len({tuple(r) for r in r4c[0, :5040].tolist()}) == 5040
    assert (np.diff(g[0, :5040]) >= 0).all()


def test_brute_force_probs(eng):
    # bruteForceProb's accumulation (no 42-gate) on the GPU with the k the reference derives, vs the checker
    frames = wl.kitti_like_frames(24, nL=6, nM=3)
    conds, nLs, ks, want = [], [], [], []
    for f in frames:
        cond, idx = ol.condition_costs(f, 9, 3)
        p, nf, uk = ol.brute_force_prob(cond, len(idx) - 3, 3)
        conds.append(cond); nLs.append(len(idx) - 3); ks.append(uk); want.append(p)
    k = max(ks)
    probs, nf = eng.weights(conds, nLs, [3] * len(conds), k, brute_force=True)
    for p, w in zip(probs, want):
        np.testing.assert_allclose(p, w, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("nw,spec", [(4, 4), (8, 6), (12, 8), (12, 12), (16, 8), (16, 16), (8, 1), (8, 3)])
def test_every_launch_shape(monkeypatch, nw, spec):
    """small batches now pick the latency shape (16 waves); every workgroup size / speculation depth the library can
    choose is pinned here against the oracle, whatever the batch size."""
    monkeypatch.setenv("KBEST_NWAVES", str(nw))
    monkeypatch.setenv("KBEST_SPEC", str(spec))
    monkeypatch.setenv("KBEST_NO_SMALL", "1")  # (problems of <= 32 rows through this kernel as well)
    e = pk.KBestEngine(0)
    rng = np.random.default_rng(100 + nw)
    for N, M, k, B in ((64, 64, 200, 6), (40, 17, 120, 5), (12, 12, 60, 9)):
        costs = rng.random((B, N * M)) * 25
        _same_as_oracle(e, costs, N, M, k, tag=(nw, spec, N, M))



@pytest.mark.parametrize("nw", [2, 4, 8, 16])
def test_every_shape_of_the_small_problem_kernel(monkeypatch, nw):
    """kbest_small.hip in every workgroup size it can be launched with (2 ... 16 waves = 4 ... 32 half-wave workers),
    whatever the batch size: square and rectangular (implicit zero columns), +inf entries, cutoff, maximise, k up to
    its limit -- nf, row4col, gains bit-for-bit, col4row up to the numbering of the padded columns."""
    monkeypatch.setenv("KBEST_SMALL_NW", str(nw))
    monkeypatch.setenv("KBEST_FORCE_SMALL", "1")
    e = pk.KBestEngine(0)
    rng = np.random.default_rng(500 + nw)
    for N, M, k, B, mode in ((32, 32, 200, 7, 0), (28, 10, 200, 40, 1), (17, 17, 64, 9, 2), (32, 5, 300, 11, 3),
                             (9, 9, 1, 5, 0), (20, 1, 30, 4, 1), (31, 30, 1000, 2, 0), (2, 2, 5, 3, 2)):
        costs = rng.random((B, N * M)) * 12
        if mode == 1:  # gate structure: sparse finite entries + one finite dummy per column
            costs[rng.random((B, N * M)) < 0.55] = np.inf
            for c in range(M):
                costs[:, c * N + (N - M + c)] = 10.0
        maximize = mode == 2
        cutoff = 42.0 if mode == 1 else (3.0 if mode == 3 else None)
        _same_as_oracle(e, costs, N, M, k, maximize, cutoff, tag=(nw, N, M, k, mode))
