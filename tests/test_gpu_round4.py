"""GPU tests added in round 4: whole result tables of ragged batches (padding and unused slots defined), the multi-device
entry's host timeline, the host entry's int8 staging, the general-size kernel beyond 512 rows."""
import numpy as np
import pytest

import oracle_lib as ol
import probabilisticsemslam_amd as pk

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def engine_with(monkeypatch, **env):
    """A fresh context whose launch knobs come from the environment at creation (kbest_create reads them once)."""
    for key, val in env.items():
        monkeypatch.setenv(key, str(val))
    eng = pk.KBestEngine(0)
    for key in env:
        monkeypatch.delenv(key)
    return eng


def _ragged(rng, B, N, M):
    nRow = rng.integers(2, N + 1, B).astype(np.int32)
    nCol = np.minimum(rng.integers(1, M + 1, B), nRow).astype(np.int32)
    nRow[0], nCol[0] = N, M  # (the maxima are reached)
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(nRow[:-1].astype(np.int64) * nCol[:-1])
    flat = rng.random(int(off[-1] + int(nRow[-1]) * int(nCol[-1])))
    return nRow, nCol, off, flat


def _expected_tables(flat, nRow, nCol, off, N, M, k, **kw):
    B = len(nRow)
    r4c = np.full((B, k, M), -1, np.int64)
    c4r = np.full((B, k, N), -1, np.int64)
    g = np.zeros((B, k))
    nf = np.zeros(B, np.int64)
    for b in range(B):
        n, m = int(nRow[b]), int(nCol[b])
        onf, or4c, oc4r, og = ol.orc_kbest(flat[off[b]: off[b] + n * m], n, m, k, **kw)
        nf[b] = onf
        r4c[b, :onf, :m] = or4c[:onf]
        oc = np.asarray(oc4r[:onf]).copy()
        oc[oc >= m] = -1  # rows on zero-padded columns
        c4r[b, :onf, :n] = oc
        g[b, :onf] = og[:onf]
    return nf, r4c, c4r, g


@pytest.mark.parametrize("knobs", [{}, {"KBEST_NO_SMALL": 1, "KBEST_NO_LANE": 1}, {"KBEST_FORCE_SMALL": 1}, {"KBEST_FORCE_WIDE": 1}])
@pytest.mark.parametrize("registered", [False, True])
def test_ragged_batch_whole_tables_are_defined(monkeypatch, knobs, registered):
    """Every entry of row4col / col4row / gain of a ragged batch has a defined value: the assignments where a problem has
    columns / rows, -1 in the padding [nCol[b], maxCol) / [nRow[b], maxRow) of emitted slots and in the slots beyond nf,
    gain 0 there -- whatever the buffers held before (recycled device blocks, the caller's registered memory)."""
    eng = engine_with(monkeypatch, **knobs)
    rng = np.random.default_rng(404)
    B, N, M, k = 37, 24, 17, 40
    nRow, nCol, off, flat = _ragged(rng, B, N, M)
    # poison what the staging buffers will be drawn from: a uniform batch of the same table sizes, all slots used
    eng.kbest(rng.random((B, N * M)), N, M, k)
    want = _expected_tables(flat, nRow, nCol, off, N, M, k)
    if registered:
        r4c = np.full((B, k, M), 0x5555, np.int32)
        c4r = np.full((B, k, N), 0x5555, np.int32)
        g = np.full((B, k), 7.0)
        nf = np.full(B, 99, np.int32)
        costs = flat.copy()
        eng.register_host(costs, r4c, c4r, g, nf)
        o = eng._opts(False, None)
        import ctypes as C
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, p(nRow), p(nCol), p(costs), p(off), k, p(r4c), p(c4r), p(g), p(nf), None)
        eng.unregister_host(costs, r4c, c4r, g, nf)
        assert rc == 0
    else:
        nf, r4c, c4r, g = eng.kbest(flat, N, M, k, nRow=nRow, nCol=nCol, costOff=off)
    assert (nf == want[0]).all()
    c = c4r.astype(np.int64).copy()
    for b in range(B):
        blk = c[b, :, : nRow[b]]
        blk[blk >= nCol[b]] = -1
    assert (r4c == want[1]).all()
    assert (c == want[2]).all()
    assert (bits(g) == bits(want[3])).all()
    eng.close()


def _multi_vs_single(eng, ids, costs, N, M, k, **kw):
    multi = pk.KBestMulti(ids)
    multi.kbest(costs, N, M, k, **kw)  # (the first call of a context creates its piece streams: not what the timeline is about)
    # The timeline is host wall-clock: one descheduled worker thread (a device's share of this batch is done in under a
    # millisecond) can spoil a single reading.  What is claimed is that the structure ALLOWS every upload to be issued before any
    # device is done -- the best of a few calls; results are checked on the last one.
    tl = None
    for _ in range(5):
        got = multi.kbest(costs, N, M, k, **kw)
        assert multi.tables_agree()
        t = multi.timeline()
        if tl is None or (t.size and t[:, 1].max() - t[:, 3].min() < tl[:, 1].max() - tl[:, 3].min()):
            tl = t
        if tl.size and tl[:, 1].max() < tl[:, 3].min():
            break
    multi.close()
    want = eng.kbest(costs, N, M, k)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (bits(got[3]) == bits(want[3])).all()
    return got, want, tl


@pytest.mark.parametrize("G", [2, 4, 8])
def test_multi_entry_logical_devices_batch_mode(engine, G):
    """kbest_create_multi with one GPU named G times: G contexts, G host workers, the slices exchanged by device-to-device
    copies.  The whole host path of the multi-device entry with G > 1 on one GPU: results equal the single-device entry's,
    every logical device holds the same global table, and the host timeline shows that every device's upload had been
    issued before ANY device was done being fed -- no device waits for another one's copies (kbest_multi.cpp)."""
    from probabilisticsemslam_amd import workloads as wl
    B, N, M, k = 515, 64, 64, 200  # (not divisible by G: the last device's padding)
    costs = wl.dense_batch(B, N, M, 0x4D554C)
    got, want, tl = _multi_vs_single(engine, [0] * G, costs, N, M, k)
    c = got[2].copy(); c[c >= M] = -1
    w = want[2].copy(); w[w >= M] = -1
    assert (c == w).all()
    assert tl.shape == (G, 6)
    assert (tl[:, 1] > 0).all() and (tl[:, 3] >= tl[:, 2]).all() and (tl[:, 5] >= tl[:, 4]).all()
    assert tl[:, 1].max() < tl[:, 3].min(), f"a device was fed before another one's upload was even issued:\n{tl}"
    # ragged shapes through the same path
    rng = np.random.default_rng(5)
    Br, Nr, Mr, kr = 67, 30, 12, 40
    costs = rng.random((Br, Nr * Mr))
    nRow = rng.integers(12, Nr + 1, Br).astype(np.int32)
    nCol = np.minimum(rng.integers(1, Mr + 1, Br), nRow).astype(np.int32)
    packed = np.zeros((Br, Nr * Mr))
    for b in range(Br):
        packed[b, : nRow[b] * nCol[b]] = rng.random(int(nRow[b]) * int(nCol[b]))
    multi = pk.KBestMulti([0] * G)
    nf, r4c, c4r, g = multi.kbest(packed, Nr, Mr, kr, nRow=nRow, nCol=nCol)
    assert multi.tables_agree()
    multi.close()
    for b in range(0, Br, 7):
        n, m = int(nRow[b]), int(nCol[b])
        onf, or4c, _, og = ol.orc_kbest(packed[b, : n * m], n, m, kr)
        assert nf[b] == onf and (r4c[b, :onf, :m] == or4c[:onf]).all() and (bits(g[b, :onf]) == bits(og[:onf])).all()


@pytest.mark.parametrize("G,S", [(2, 2), (3, 8), (4, 4)])
def test_multi_entry_logical_devices_subtree_mode(engine, G, S):
    """Subtree mode over logical devices: every device enumerates its shards of every matrix, one exchange, the merge on every
    device, each device sends its share of the merged table home; equal to the single-device result."""
    from probabilisticsemslam_amd import workloads as wl
    B, N, M, k = 9, 48, 48, 120
    costs = wl.dense_batch(B, N, M, 0x535542)
    multi = pk.KBestMulti([0] * G)
    got = multi.kbest(costs, N, M, k, subtree=True, n_shard=S)
    assert multi.tables_agree()
    tl = multi.timeline()
    multi.close()
    want = engine.kbest(costs, N, M, k)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (bits(got[3]) == bits(want[3])).all()
    assert (tl[:, 5] >= tl[:, 4]).all() and (tl[:, 1] > 0).all()


def test_subtree_shards_from_differently_configured_contexts(monkeypatch):
    """Root-subtree sharding partitions on the REFERENCE's column (kbest_c.h), not on a position in the enumeration's own
    column order: shards enumerated by contexts that differ in kernel, launch shape and knobs still form disjoint, complete
    partitions, and their merge is the global k best."""
    import torch
    from probabilisticsemslam_amd import distributed as kd
    rng = np.random.default_rng(77)
    B, N, M, k, S = 6, 40, 40, 90, 3
    costs = rng.random((B, N * M))
    knobs = [{}, {"KBEST_NO_REORDER": 1, "KBEST_NWAVES": 8, "KBEST_SPEC": 6}, {"KBEST_FORCE_WIDE": 1}]
    G, R, F = [], [], []
    for s in range(S):
        eng = engine_with(monkeypatch, **knobs[s])
        nf, r4c, _, g = eng.kbest(costs, N, M, k, root_shard=(s, S))
        G.append(torch.from_numpy(g)); R.append(torch.from_numpy(r4c.astype(np.int64))); F.append(torch.from_numpy(nf.astype(np.int64)))
        eng.close()
    mg, mr, mnf = kd.merge_lists(torch.stack(G), torch.stack(R), torch.stack(F), k)
    onf, or4c, _, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (mnf.numpy() == onf).all() and (mr.numpy() == or4c).all() and (bits(mg.numpy()) == bits(og)).all()


def test_graph_capture_of_the_device_entry(engine):
    """kbest_batch_f64_dev is legal inside a stream capture (kbest_c.h): no event of the context's cross-stream bookkeeping may
    end up in the graph.  Captured once, replayed twice on new inputs; then a plain launch on another stream still works."""
    import torch
    from probabilisticsemslam_amd import workloads as wl
    B, N, M, k = 300, 16, 16, 30
    dev = torch.device("cuda", 0)
    costs = wl.dense_batch(B, N, M, 0x47524150)
    d_cost = torch.from_numpy(costs).to(dev)
    d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
    d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_nf = torch.empty(B, dtype=torch.int32, device=dev)
    engine.reserve(B, N, k)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # once outside the capture: lazy one-off work of the runtime (function attributes)
        engine.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_g, d_nf, stream=s.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        engine.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_g, d_nf, stream=s.cuda_stream)
    for seed in (1, 2):
        costs = wl.dense_batch(B, N, M, 0x47524150 + seed)
        d_cost.copy_(torch.from_numpy(costs))
        d_g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        onf, or4c, _, og, _ = ol.orc_kbest_batch(costs[:40], N, M, k)
        assert (d_nf.cpu().numpy()[:40] == onf).all() and (d_r4c.cpu().numpy()[:40] == or4c).all()
        assert (bits(d_g.cpu().numpy()[:40]) == bits(og)).all()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        engine.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_g, d_nf, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert (bits(d_g.cpu().numpy()[:40]) == bits(og)).all()


@pytest.mark.parametrize("shape", [(1024, 1024, 10, 1), (700, 300, 12, 2), (513, 513, 6, 2), (600, 40, 30, 3)])
def test_general_size_kernel_beyond_512_rows(engine, shape):
    """513 ... 1 024 rows: sixteen rows per lane, four waves per problem (kbest_wide.hip) -- kBest2D takes any numRow >= numCol
    (shortestPathCPP.cpp:571-644); bit for bit against the checker."""
    N, M, k, B = shape
    rng = np.random.default_rng(N + M)
    costs = rng.random((B, N * M))
    nf, r4c, c4r, g = engine.kbest(costs, N, M, k)
    for b in range(B):
        onf, or4c, oc4r, og = ol.orc_kbest(costs[b], N, M, k)
        assert nf[b] == onf and (r4c[b, :onf] == or4c[:onf]).all() and (bits(g[b, :onf]) == bits(og[:onf])).all()
        c = c4r[b, :onf].copy(); c[c >= M] = -1
        w = np.asarray(oc4r[:onf]).copy(); w[w >= M] = -1
        assert (c == w).all()


@pytest.mark.parametrize("shape", [(1027, 64, 200), (300, 32, 200), (3000, 5, 200), (2100, 16, 150)])
def test_narrow_staging_of_the_host_entry_equals_the_wide_tables(monkeypatch, shape):
    """kbest_batch_f64 on uniform square batches: the kernels write row4col as bytes into pinned staging and host threads widen
    it into the caller's int32 row4col and its inverse col4row, piece by piece.  Every table equals what the same entry gives
    with the int32 tables written by the kernel itself (KBEST_NO_NARROW) -- also with fewer than k solutions (5x5: 120
    assignments), with forbidden arcs, and from six host threads at once on one context."""
    import threading
    B, N, k = shape
    rng = np.random.default_rng(B + N)
    costs = rng.random((B, N * N))
    if N == 16:
        costs[rng.random(costs.shape) < 0.3] = np.inf
    eng = pk.KBestEngine(0)
    wide = engine_with(monkeypatch, KBEST_NO_NARROW=1)
    a = eng.kbest(costs, N, N, k)
    b = wide.kbest(costs, N, N, k)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all() and (bits(a[3]) == bits(b[3])).all()
    if N == 5:
        assert (a[0] == 120).all() and (a[1][:, 120:] == -1).all() and (a[2][:, 120:] == -1).all() and (a[3][:, 120:] == 0).all()
    for bb in range(0, B, max(1, B // 5)):
        onf, or4c, oc4r, og = ol.orc_kbest(costs[bb], N, N, k)
        assert a[0][bb] == onf and (a[1][bb, :onf] == or4c[:onf]).all() and (a[2][bb, :onf] == oc4r[:onf]).all()
    if N == 32:
        out, errs = [None] * 6, []

        def work(i):
            try:
                out[i] = eng.kbest(costs, N, N, k)
            except Exception as ex:  # noqa: BLE001
                errs.append(ex)
        th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs
        for o in out:
            assert (o[0] == a[0]).all() and (o[1] == a[1]).all() and (o[2] == a[2]).all() and (bits(o[3]) == bits(a[3])).all()
    eng.close()
    wide.close()


@pytest.mark.parametrize("knobs", [{"KBEST_OPT_RHO0": 0.85}, {"KBEST_OPT_RHO0": 0.4, "KBEST_OPT_KAPPA": 0.05}, {"KBEST_OPT_RHO0": 0.6, "KBEST_SPEC": 2},
                                   {"KBEST_OPT_RHO0": 0.3, "KBEST_OPT_MINPOOL": 2}])
def test_optimistic_bounds_keep_the_enumeration_exact(monkeypatch, knobs):
    """Optimistic bounds with re-split tickets (kbest_engine.hip, struct Opt), forced on everywhere they compile in (4-wave 64-row
    kernel) at quantiles from the tuned one down to absurdly tight ones -- many tickets, nodes split three and four times, tiny
    steps between re-splits --: a bounded slice of the randomised soak (seven cost structures incl. exact ties and near-ties at
    1e-9, +inf patterns, rectangular, maximise, cutoff, k = 1 ... 300) and the dense 32x32, k = 200 case the default routing
    uses them for.  Whatever the guess, results are the checker's bit for bit."""
    import soak_lib
    from probabilisticsemslam_amd import workloads as wl
    env = {"KBEST_NO_SMALL": 1, "KBEST_NO_LANE": 1, "KBEST_NWAVES": 4, "KBEST_SPEC": 4}
    env.update(knobs)
    eng = engine_with(monkeypatch, **env)
    ncase, nprob, bad = soak_lib.run(eng, seed=20261004, n_cases=250, big_frac=0.0, big_max=64)
    assert bad is None, bad
    assert ncase >= 60
    costs, N, M, k = wl.dense_config("c3", B=96)
    nf, r4c, c4r, g = eng.kbest(costs, N, M, k)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (nf == onf).all() and (r4c == or4c).all() and (c4r == oc4r).all() and (bits(g) == bits(og)).all()
    eng.close()


def _assoc_expected(frame, nL, nM, k):
    """getAssignmentProbs on one raw block by the checker: conditionCosts -> assignmentProb -> scatter back (assignment.cpp:38-74)."""
    cond, idx = ol.condition_costs(frame, nL + nM, nM)
    cl = len(idx) - nM
    po, nf = ol.assignment_prob(cond, cl, nM, k)
    want = np.zeros((nM, nL + 1))
    want[:, idx[:cl]] = po[:, :cl]
    want[:, nL] = po[:, cl]
    return want, nf


TINY_SHAPES = [(6, 3), (6, 5), (4, 2), (9, 4), (0, 3), (12, 3), (3, 6), (0, 8), (5, 5), (40, 2), (62, 2), (1, 2), (20, 5), (30, 4), (12, 5)]


@pytest.mark.parametrize("k", [200, 7])
def test_frames_with_a_handful_of_measurements_by_exhaustive_enumeration(monkeypatch, k):
    """kbest_tiny.hip: frames whose assignments are few in all ((nL + nM)! / nL! <= 2^23: the reference's real frame sizes,
    README.md:11) are answered by looking at every assignment instead of enumerating the k best.  Same probabilities as the
    checker's getAssignmentProbs chain, and bit for bit those of the enumeration kernel on the same frames (same gains, same
    order of additions) -- frame by frame (one frame per call: the reference's call pattern), as one mixed batch, and as a batch
    that fills the chip (smaller workgroups)."""
    from probabilisticsemslam_amd import workloads as wl
    tiny = pk.KBestEngine(0)
    plain = engine_with(monkeypatch, KBEST_NO_TINY=1)
    frames, nLs, nMs = [], [], []
    for i, (nL, nM) in enumerate(TINY_SHAPES):
        for f in wl.kitti_like_frames(3, nL=nL, nM=nM, seed=0x7151 + 97 * i) if nL > 0 else [None] * 2:
            if f is None:  # no landmarks: every measurement is new (assignment.cpp:52-54)
                f = np.full(nM * nM, np.inf)
                for c in range(nM):
                    f[c * nM + c] = 10.0
            frames.append(f)
            nLs.append(nL)
            nMs.append(nM)
    # one frame per call
    for f, nL, nM in zip(frames, nLs, nMs):
        out, nf = tiny.weights([f], [nL], [nM], k, condition=True)
        ref, nfr = plain.weights([f], [nL], [nM], k, condition=True)
        want, nfw = _assoc_expected(f, nL, nM, k)
        assert nf[0] == nfr[0] == nfw, (nL, nM, nf, nfr, nfw)
        assert np.array_equal(out[0], ref[0]), (nL, nM)
        np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-12)
    # one mixed batch, and a batch of 600 frames (256-thread workgroups)
    out, nf = tiny.weights(frames, nLs, nMs, k, condition=True)
    ref, nfr = plain.weights(frames, nLs, nMs, k, condition=True)
    assert (nf == nfr).all()
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)
    # assignmentProb alone, on blocks that are conditioned already (the reference's own call, assignment.cpp:58-62; the shim's
    # path): the exhaustive kernel takes them too; a block that is NOT a conditioned one (negative entries, no exact zero) is
    # handed back and answered by the enumeration kernel
    conds, cls, cms = [], [], []
    for f, nL, nM in zip(frames, nLs, nMs):
        if nL == 0:
            continue
        c, idx = ol.condition_costs(f, nL + nM, nM)
        conds.append(c)
        cls.append(len(idx) - nM)
        cms.append(nM)
    out, nf = tiny.weights(conds, cls, cms, k)
    ref, nfr = plain.weights(conds, cls, cms, k)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    for c, cl, cm, o in zip(conds[:6], cls, cms, out):
        po, _ = ol.assignment_prob(c, cl, cm, k)
        np.testing.assert_allclose(o, po, rtol=0, atol=1e-12)
    rng = np.random.default_rng(11)
    odd = [rng.normal(size=(7 + 3) * 3) * 5.0 for _ in range(4)]
    out, nf = tiny.weights(odd, [7] * 4, [3] * 4, k)
    ref, nfr = plain.weights(odd, [7] * 4, [3] * 4, k)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    many = wl.kitti_like_frames(600, nL=6, nM=4, seed=0xBEEF)
    out, nf = tiny.weights(many, [6] * 600, [4] * 600, k, condition=True)
    ref, nfr = plain.weights(many, [6] * 600, [4] * 600, k, condition=True)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    tiny.close()
    plain.close()


def test_exhaustive_enumeration_with_ties_and_with_too_many_of_them(monkeypatch):
    """Integer costs: masses of exact ties.  With k beyond the number of assignments the order of equal gains cannot matter and
    the probabilities are the checker's.  A frame whose k-th gain sits in a group of thousands of equal ones overflows the
    candidate list: it comes back through the enumeration kernels (nf = -2 inside), with valid probabilities."""
    eng = pk.KBestEngine(0)
    nL, nM = 4, 3  # 7 x 3: 210 assignments
    nR = nL + nM
    C = np.full(nR * nM, np.inf)
    for c in range(nM):
        for r in range(nL):
            C[c * nR + r] = float((r + c) % 3)
        C[c * nR + nL + c] = 2.0
    out, nf = eng.weights([C], [nL], [nM], 1024, condition=True)
    want, nfw = _assoc_expected(C, nL, nM, 1024)
    assert nfw < 1024 and nf[0] == nfw  # everything within the gate is enumerated
    np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-12)
    # 12 x 4, all costs equal: 11 880 assignments with the same gain
    Z = np.zeros(12 * 4)
    out, nf = eng.weights([Z], [8], [4], 200, condition=True)
    assert nf[0] == 200
    np.testing.assert_allclose(out[0].sum(axis=1), np.ones(4), rtol=0, atol=1e-12)
    assert (out[0] >= 0).all()
    eng.close()


BNB_SHAPES = [(20, 10), (40, 12), (30, 16), (50, 8), (25, 6), (60, 4), (10, 9), (0, 10)]


@pytest.mark.parametrize("k", [200, 7, 500])
def test_frame_sized_blocks_by_the_bounded_walk(monkeypatch, k):
    """kbest_bnb.hip: frame-sized association problems (up to 16 measurements, 64 rows) are answered by walking every assignment
    whose partial sums stay below a bound that is raised until k assignments lie below it -- no enumeration.  Same counts and,
    bit for bit, the same probabilities as the enumeration kernels on the same frames (same gains: calcGain's sums; same order
    of additions in the weights), and the checker's getAssignmentProbs chain within 1e-12: one frame per call (the reference's
    call pattern), one mixed batch, a batch that fills the chip (small workgroups), and assignmentProb alone on conditioned
    blocks."""
    from probabilisticsemslam_amd import workloads as wl
    fast = pk.KBestEngine(0)
    plain = engine_with(monkeypatch, KBEST_NO_TINY=1, KBEST_NO_BNB=1)
    frames, nLs, nMs = [], [], []
    for i, (nL, nM) in enumerate(BNB_SHAPES):
        for f in wl.kitti_like_frames(3, nL=nL, nM=nM, seed=0xB0B + 31 * i) if nL > 0 else [None] * 2:
            if f is None:
                f = np.full(nM * nM, np.inf)
                for c in range(nM):
                    f[c * nM + c] = 10.0
            frames.append(f)
            nLs.append(nL)
            nMs.append(nM)
    for j, (f, nL, nM) in enumerate(zip(frames, nLs, nMs)):
        out, nf = fast.weights([f], [nL], [nM], k, condition=True)
        ref, nfr = plain.weights([f], [nL], [nM], k, condition=True)
        assert nf[0] == nfr[0], (nL, nM, nf, nfr)
        assert np.array_equal(out[0], ref[0]), (nL, nM)
        if j % 3 == 0:
            want, nfw = _assoc_expected(f, nL, nM, k)
            assert nf[0] == nfw
            np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-12)
    out, nf = fast.weights(frames, nLs, nMs, k, condition=True)
    ref, nfr = plain.weights(frames, nLs, nMs, k, condition=True)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    many = wl.kitti_like_frames(400, nL=20, nM=10, seed=0xFACE)
    out, nf = fast.weights(many, [20] * 400, [10] * 400, k, condition=True)
    ref, nfr = plain.weights(many, [20] * 400, [10] * 400, k, condition=True)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    conds, cls, cms = [], [], []
    for f, nL, nM in zip(frames, nLs, nMs):
        if nL == 0:
            continue
        c, idx = ol.condition_costs(f, nL + nM, nM)
        conds.append(c)
        cls.append(len(idx) - nM)
        cms.append(nM)
    out, nf = fast.weights(conds, cls, cms, k)
    ref, nfr = plain.weights(conds, cls, cms, k)
    assert (nf == nfr).all() and all(np.array_equal(a, b) for a, b in zip(out, ref))
    fast.close()
    plain.close()


def test_bounded_walk_hands_back_what_it_cannot_bound(monkeypatch):
    """All costs equal: every one of the 20!/10! assignments has the same gain -- no bound separates k of them from the rest.  The
    frame comes back through the enumeration kernels (nf = -2 inside) with valid probabilities; its neighbours in the batch are
    answered by the walk."""
    from probabilisticsemslam_amd import workloads as wl
    eng = pk.KBestEngine(0)
    fr = wl.kitti_like_frames(3, nL=10, nM=10, seed=5)
    flat = np.full(20 * 10, np.inf)
    for c in range(10):
        flat[c * 20: c * 20 + 10] = 1.0
        flat[c * 20 + 10 + c] = 1.0
    batch = [fr[0], flat, fr[1], fr[2]]
    out, nf = eng.weights(batch, [10] * 4, [10] * 4, 200, condition=True)
    assert nf[1] == 200
    np.testing.assert_allclose(out[1].sum(axis=1), np.ones(10), rtol=0, atol=1e-12)
    for i, f in ((0, fr[0]), (2, fr[1]), (3, fr[2])):
        want, nfw = _assoc_expected(f, 10, 10, 200)
        assert nf[i] == nfw
        np.testing.assert_allclose(out[i], want, rtol=0, atol=1e-12)
    # the device-pointer entry stays total and asynchronous: a second launch answers the frame that was handed back
    import torch
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    F, nL, nM, nR, k = 4, 10, 10, 20, 200
    d_probs = torch.zeros(F * nM * (nL + 1), dtype=torch.float64, device=dev)
    d_nf = torch.zeros(F, dtype=torch.int32, device=dev)
    eng.reserve_assoc(F, nR, nM, k)
    eng.assoc_probs_dev(F, nR, nM, t(np.full(F, nL, np.int32)), t(np.full(F, nM, np.int32)), t(np.full(F, nR, np.int32)), t(np.concatenate(batch)),
                        t(np.arange(F, dtype=np.int64) * nR * nM), k, d_probs, t(np.arange(F, dtype=np.int64) * nM * (nL + 1)), d_nf,
                        stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_nf.cpu().numpy() == nf).all()
    dp = d_probs.cpu().numpy().reshape(F, nM, nL + 1)
    for i in (0, 2, 3):
        assert np.array_equal(dp[i], out[i])
    np.testing.assert_allclose(dp[1].sum(axis=1), np.ones(10), rtol=0, atol=1e-12)
    eng.close()
