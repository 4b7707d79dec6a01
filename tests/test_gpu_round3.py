"""GPU tests added in round 3: the lane-per-child kernel (kbest_lane.hip) in every launch shape, a bounded fixed-seed soak
over all cost structures on every kernel, reservations that cover smaller batches, the chunked host entry on ragged
batches."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import probabilisticsemslam_amd as pk
import soak_lib
from probabilisticsemslam_amd import workloads as wl

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


def canon(c4r, M):
    c = c4r.copy()
    c[c >= M] = -1  # rows on zero-padded columns: which padded column is a tie artefact
    return c


LANE_CASES = [(16, 16, 50, 40, {}), (32, 32, 200, 12, {}), (8, 8, 10, 30, {}), (20, 20, 64, 16, {}), (32, 32, 7, 20, {}),
              (24, 10, 100, 20, {}), (16, 5, 30, 20, {}), (30, 30, 200, 8, {"maximize": True}), (12, 12, 300, 10, {}),
              (32, 20, 150, 10, {"cutoff": 0.3}), (16, 16, 50, 16, {"cutoff": 0.05}), (5, 5, 200, 10, {}), (1, 1, 3, 4, {}),
              (3, 2, 9, 6, {}), (17, 17, 40, 9, {}), (16, 16, 1, 5, {})]


@pytest.mark.parametrize("nw,spec", [(1, 1), (2, 3), (4, 6), (1, 8), (2, 6), (1, 3), (4, 8), (2, 16)])
def test_lane_kernel_every_launch_shape(monkeypatch, nw, spec):
    """kbest_lane.hip forced for every plain batch of <= 32-row problems: waves per problem, hypotheses per round swept; square, rectangular, maximise, cutoff, exhaustive (k beyond the number of assignments), exact ties
    (multisets), k = 1.  nf, row4col and gains bit for bit against the oracle, col4row after mapping padded columns."""
    eng = engine_with(monkeypatch, KBEST_FORCE_LANE=1, KBEST_LANE_NW=nw, KBEST_LANE_SPEC=spec)
    rng = np.random.default_rng(100 * nw + spec)
    for (N, M, k, B, kw) in LANE_CASES:
        costs = rng.random((B, N * M))
        ties = N == 12
        if ties:
            costs = np.floor(costs * 4)
        nf, r4c, c4r, g = eng.kbest(costs, N, M, k, **kw)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, **kw)
        assert (nf == onf).all(), (N, M, k, kw)
        for b in range(B):
            n = int(nf[b])
            if ties:
                assert (np.sort(g[b, :n]) == np.sort(og[b, :n])).all()
                continue
            assert (r4c[b, :n] == or4c[b, :n]).all(), (N, M, k, kw, b)
            assert (bits(g[b, :n]) == bits(og[b, :n])).all(), (N, M, k, kw, b)
            assert (canon(c4r[b, :n], M) == canon(oc4r[b, :n], M)).all(), (N, M, k, kw, b)
    eng.close()


def test_lane_kernel_takes_the_dense_16_row_batches(monkeypatch):
    """Default routing: a chip-filling batch of dense 16x16 problems runs on the lane-per-child kernel and equals both the
    64-row kernel's result (KBEST_NO_LANE) and the oracle; ragged shapes inside one launch as well."""
    costs, N, M, k = wl.dense_config("c2")
    eng = pk.KBestEngine(0)
    old = engine_with(monkeypatch, KBEST_NO_LANE=1)
    a = eng.kbest(costs, N, M, k)
    b = old.kbest(costs, N, M, k)
    for x, y in zip(a, b):
        assert (np.asarray(x) == np.asarray(y)).all()
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (a[0] == onf).all() and (a[1] == or4c).all() and (bits(a[3]) == bits(og)).all()
    # ragged: per-problem shapes up to 16 x 16, packed offsets
    rng = np.random.default_rng(3)
    B = 700
    nRow = rng.integers(2, 17, B).astype(np.int32)
    nCol = np.array([rng.integers(1, r + 1) for r in nRow], dtype=np.int32)
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(nRow[:-1].astype(np.int64) * nCol[:-1])
    flat = rng.random(int(off[-1] + nRow[-1] * nCol[-1]))
    lane = engine_with(monkeypatch, KBEST_FORCE_LANE=1)
    nf, r4c, c4r, g = lane.kbest(flat, 16, 16, 20, nRow=nRow, nCol=nCol, costOff=off)
    for i in range(0, B, 7):
        n, m = int(nRow[i]), int(nCol[i])
        onf1, or4c1, _, og1 = ol.orc_kbest(flat[off[i]: off[i] + n * m], n, m, 20)
        assert nf[i] == onf1 and (r4c[i, :onf1, :m] == or4c1[:onf1]).all() and (bits(g[i, :onf1]) == bits(og1[:onf1])).all()
    for e in (eng, old, lane):
        e.close()


@pytest.mark.parametrize("knobs", [{}, {"KBEST_FORCE_LANE": 1}, {"KBEST_FORCE_SMALL": 1}, {"KBEST_NO_SMALL": 1, "KBEST_NO_LANE": 1},
                                   {"KBEST_FORCE_WIDE": 1}])
def test_fixed_seed_soak_on_every_kernel(monkeypatch, knobs):
    """A bounded slice of the randomised soak (tests/soak_lib.py: seven cost structures incl. exact ties, +inf patterns,
    near-ties at 1e-9; rectangular and square; maximise; cutoff; k = 1 ... 300), fixed seed, on the default routing and with
    each kernel forced (the a-priori thresholds are on wherever the launch shape has them)."""
    eng = engine_with(monkeypatch, **knobs)
    ncase, nprob, bad = soak_lib.run(eng, seed=20261003, n_cases=600 if not knobs else 300, big_frac=0.05 if not knobs else 0.0,
                                     big_max=130)
    eng.close()
    assert bad is None, bad
    assert ncase >= 60


def test_soak_large_batches_with_thresholds_on(engine):
    """The a-priori thresholds (T0 / T1) only run in the 8+ wave launch shapes, i.e. for batches that fill the chip: the
    same cost structures in batches of 300 ... 1200 problems of 33 ... 64 rows."""
    rng = np.random.default_rng(77)
    for _ in range(6):
        case = soak_lib.draw_case(rng, max_rows=64, batches=(300, 700, 1200))
        while case["N"] < 33 or case["k"] < 3:
            case = soak_lib.draw_case(rng, max_rows=64, batches=(300, 700, 1200))
        bad = soak_lib.check_case(engine, case)
        assert bad is None, bad


def test_reservation_covers_smaller_batches():
    """kbest_c.h: after kbest_reserve(B, ...) 'launches of up to B problems' never allocate.  A smaller batch may pick
    another kernel or launch shape (more waves, more state slots per problem): every tier has to be covered."""
    import torch
    dev = torch.device("cuda", 0)
    eng = pk.KBestEngine(0)
    for (N, k, Bres) in ((32, 200, 1100), (16, 50, 2000), (28, 64, 600)):
        eng.reserve(Bres, N, k)
        for B in (Bres, 1025, 600, 513, 512, 300, 257, 256, 100, 3, 1):
            if B > Bres:
                continue
            for M in (N, max(1, N // 3)):
                costs = wl.dense_batch(B, N, M, 5 + B + M)
                d_cost = torch.from_numpy(costs).to(dev)
                r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
                c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
                g = torch.empty((B, k), dtype=torch.float64, device=dev)
                nf = torch.empty(B, dtype=torch.int32, device=dev)
                eng.kbest_dev(d_cost, B, N, M, k, r4c, c4r, g, nf)  # raises KBestError(NOT_RESERVED) if the reservation is short
                torch.cuda.synchronize()
                b = B // 2
                onf, or4c, _, og = ol.orc_kbest(costs[b], N, M, k)
                assert int(nf[b]) == onf and (r4c[b, :onf].cpu().numpy() == or4c[:onf]).all()
                assert (bits(g[b, :onf].cpu().numpy()) == bits(og[:onf])).all()
    eng.close()


def test_chunked_host_entry_odd_ragged_and_pushed(engine):
    """kbest_batch_f64 sends a large batch through the GPU in chunks (upload / kernel / copy back overlap): odd batch sizes,
    per-problem shapes with packed offsets, and the push counter go through the same path."""
    rng = np.random.default_rng(9)
    # uniform, odd B, large enough outputs to be chunked
    B, N, M, k = 1027, 64, 64, 200
    costs = wl.dense_batch(B, N, M, 0xABCD)
    nf, r4c, c4r, g = engine.kbest(costs, N, M, k)
    for b in (0, 1, 513, 514, 1026):
        onf, or4c, oc4r, og = ol.orc_kbest(costs[b], N, M, k)
        assert nf[b] == onf and (r4c[b] == or4c).all() and (bits(g[b]) == bits(og)).all()
        assert (canon(c4r[b], M) == canon(oc4r, M)).all()
    # ragged + offsets
    B, N, M, k = 1501, 40, 30, 120
    nRow = rng.integers(5, N + 1, B).astype(np.int32)
    nCol = np.minimum(rng.integers(1, M + 1, B), nRow).astype(np.int32)
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum(nRow[:-1].astype(np.int64) * nCol[:-1])
    flat = rng.random(int(off[-1] + nRow[-1] * nCol[-1]))
    nf, r4c, c4r, g = engine.kbest(flat, N, M, k, nRow=nRow, nCol=nCol, costOff=off)
    for b in list(range(0, B, 97)) + [B - 1]:
        n, m = int(nRow[b]), int(nCol[b])
        onf, or4c, _, og = ol.orc_kbest(flat[off[b]: off[b] + n * m], n, m, k)
        assert nf[b] == onf and (r4c[b, :onf, :m] == or4c[:onf]).all() and (bits(g[b, :onf]) == bits(og[:onf])).all()
    # push counting (the reference's order of splits, no pruning) on a chunked batch
    B, N, M, k = 1100, 48, 48, 200
    costs = wl.dense_batch(B, N, M, 0xBEEF)
    nf, r4c, c4r, g, pushed = engine.kbest(costs, N, M, k, count_pushed=True, prune=False)
    for b in (0, 549, 550, 1099):
        onf, or4c, _, og, op = ol.orc_kbest_batch(costs[b:b + 1], N, M, k)
        assert nf[b] == onf[0] and (r4c[b] == or4c[0]).all() and pushed[b] == op[0]


def _shard_lists(eng, costs, N, M, k, S, **kw):
    """The k best of every shard's root subtrees (engine, root_shard = (s, S)), stacked [S, B, ...]."""
    out = [eng.kbest(costs, N, M, k, root_shard=(s, S), **kw) for s in range(S)]
    return (np.stack([o[3] for o in out]), np.stack([o[1] for o in out]), np.stack([o[0] for o in out]))


@pytest.mark.parametrize("S", [1, 2, 3, 8])
def test_device_merge_is_the_global_kbest(engine, S):
    """kbest_merge_topk_f64_dev (the global k-best heap of the subtree-sharded enumeration, on the device): S shards'
    lists -> the single-enumeration result = the oracle; equal to the torch merge of distributed.py (same tie rule); with
    exact ties (integer costs) the gains agree as multisets and the table does not depend on the number of shards."""
    import torch
    from probabilisticsemslam_amd import distributed as kd
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(40 + S)
    for (N, M, k, B, ties, maximize) in ((16, 16, 50, 6, False, False), (40, 12, 120, 3, False, False), (10, 10, 64, 5, True, False),
                                          (24, 24, 30, 4, False, True), (5, 5, 200, 3, False, False)):
        costs = rng.random((B, N * M))
        if ties:
            costs = np.floor(costs * 3)
        G, R, Nf = _shard_lists(engine, costs, N, M, k, S, maximize=maximize)
        tg, tr, tn = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (G, R, Nf))
        og = torch.zeros((B, k), dtype=torch.float64, device=dev)
        orr = torch.full((B, k, M), -1, dtype=torch.int32, device=dev)
        on = torch.zeros(B, dtype=torch.int32, device=dev)
        # plain [S][B][...] arrays: the three tables have different shard strides, so merge shard by shard layout = packed slices
        up16 = lambda x: (x + 15) & ~15  # noqa: E731
        off_r, off_n = up16(B * k * 8), up16(B * k * 8) + up16(B * k * M * 4)
        sl = off_n + up16(B * 4)
        pack = torch.zeros(S * sl, dtype=torch.uint8, device=dev)
        for s in range(S):
            base = s * sl
            pack[base: base + B * k * 8] = tg[s].contiguous().view(torch.uint8).reshape(-1)
            pack[base + off_r: base + off_r + B * k * M * 4] = tr[s].contiguous().view(torch.uint8).reshape(-1)
            pack[base + off_n: base + off_n + B * 4] = tn[s].contiguous().view(torch.uint8).reshape(-1)
        p0 = pack.data_ptr()
        engine.merge_topk_dev(B, S, k, M, p0, p0 + off_r, p0 + off_n, sl, og, orr, on, maximize=maximize)
        torch.cuda.synchronize()
        mg, mr, mn = og.cpu().numpy(), orr.cpu().numpy(), on.cpu().numpy()
        # the torch merge (gloo path) gives the same table, ties included
        cg, cr, cn = kd.merge_lists(tg.cpu(), tr.cpu(), tn.cpu(), k, maximize)
        for b in range(B):
            n = int(mn[b])
            assert n == int(cn[b])
            assert (bits(mg[b, :n]) == bits(cg[b, :n].numpy())).all() and (mr[b, :n] == cr[b, :n].numpy()).all()
        onf, or4c, _, ogain, _ = ol.orc_kbest_batch(costs, N, M, k, maximize=maximize)
        for b in range(B):
            n = int(onf[b])
            assert mn[b] == n
            if ties:
                assert (np.sort(mg[b, :n]) == np.sort(ogain[b, :n])).all()
                assert len({tuple(x) for x in mr[b, :n].tolist()}) == n
            else:
                assert (bits(mg[b, :n]) == bits(ogain[b, :n])).all() and (mr[b, :n] == or4c[b, :n]).all()


def test_multi_entry_subtree_mode_on_one_device(engine):
    """kbest_batch_f64_multi_ex(KBEST_MULTI_SUBTREE) with nDev = 1 standing in for 1, 2, 4 and 8 shards: enumeration per
    shard, ONE packed all-gather on the RCCL communicator, merge on the device = the batch mode = the oracle."""
    m = pk.KBestMulti([0])
    rng = np.random.default_rng(12)
    for (N, M, k, B) in ((64, 64, 200, 3), (30, 10, 100, 4), (16, 16, 50, 7)):
        costs = rng.random((B, N * M)) * 3
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
        for S in (1, 2, 4, 8):
            nf, r4c, c4r, g = m.kbest(costs, N, M, k, subtree=True, n_shard=S)
            assert (nf == onf).all() and (r4c == or4c).all() and (bits(g) == bits(og)).all(), (N, M, k, S)
            assert m.tables_agree()
            # col4row is the inverse of row4col, -1 for rows without a real column
            for b in range(B):
                for s in range(0, int(nf[b]), 17):
                    want = np.full(N, -1)
                    want[r4c[b, s]] = np.arange(M)
                    assert (c4r[b, s] == want).all()
    # bad shapes are argument errors, not kernel errors
    with pytest.raises(pk.KBestError):
        m.kbest(np.zeros((2, 12)), 4, 3, 5, nRow=[4, 2], nCol=[3, 3])
    m.close()


def test_multi_entry_two_devices():
    """The multi-device entries on two (or more) GPUs of the box: batch mode with B not divisible by the devices and with
    B smaller than the number of devices, subtree mode.  Skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
    lib = pk.load_library()
    n = lib.kbest_device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    devs = list(range(min(n, 8)))
    m = pk.KBestMulti(devs)
    rng = np.random.default_rng(2)
    for (N, M, k, B) in ((32, 32, 100, 2 * len(devs) + 1), (16, 16, 50, 1), (64, 64, 200, len(devs) - 1)):
        costs = rng.random((B, N * M))
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
        nf, r4c, c4r, g = m.kbest(costs, N, M, k)
        assert (nf == onf).all() and (r4c == or4c).all() and (bits(g) == bits(og)).all()
        assert m.tables_agree()
        nf, r4c, c4r, g = m.kbest(costs, N, M, k, subtree=True)
        assert (nf == onf).all() and (r4c == or4c).all() and (bits(g) == bits(og)).all()
        assert m.tables_agree()
    m.close()


def test_registered_host_buffers_take_the_direct_path(engine):
    """kbest_register_host_buffer: result tables in registered host memory are written by the kernels themselves (no staging,
    no copy back), cost blocks come up asynchronously; results identical to the copying path, slots beyond nf get the same
    defined values, partly registered argument sets fall back to the copying path."""
    import ctypes as C
    rng = np.random.default_rng(21)
    for (N, M, k, B, kw) in ((64, 64, 200, 300, {}), (16, 16, 50, 700, {}), (30, 10, 100, 40, {"cutoff": 1.0}), (5, 5, 200, 9, {}),
                             (100, 20, 30, 12, {})):
        costs = np.ascontiguousarray(rng.random((B, N * M)) * 4)
        want = engine.kbest(costs, N, M, k, **kw)
        r4c = np.full((B, k, M), 7, np.int32); c4r = np.full((B, k, N), 7, np.int32)
        gain = np.full((B, k), 7.0); nf = np.full(B, 7, np.int32)
        o = engine._opts(False, kw.get("cutoff"))
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        engine.register_host(costs, r4c, c4r, gain, nf)
        rc = engine.lib.kbest_batch_f64(engine.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r), p(gain), p(nf), None)
        engine.unregister_host(costs, r4c, c4r, gain, nf)
        assert rc == 0
        assert (nf == want[0]).all() and (r4c == want[1]).all() and (c4r == want[2]).all() and (bits(gain) == bits(want[3])).all()
    # only some of the buffers registered: the copying path, same results
    engine.register_host(r4c)
    got = engine.kbest(costs, N, M, k)
    engine.unregister_host(r4c)
    assert all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(got, want))


def test_one_matrix_over_several_workgroups(monkeypatch):
    """(KBEST_SPLIT: measured slower than the unsplit launch and therefore off by default -- kbest_capi.cpp, split_factor --, but
    kept exact.)  A batch of 33 ... 64-row square problems that leaves CUs idle is enumerated by 2 or 4 workgroups per matrix (the root's
    subtrees in turn, thresholds shared through HBM) and merged on the device: results are the unsplit launch's and the
    oracle's bit for bit, col4row is the inverse of row4col; maximise and cutoff go through the same path."""
    import torch
    dev = torch.device("cuda", 0)
    plain = engine_with(monkeypatch, KBEST_NO_SPLIT=1)
    engs = {"four": engine_with(monkeypatch, KBEST_SPLIT=4), "two": engine_with(monkeypatch, KBEST_SPLIT=2)}
    rng = np.random.default_rng(61)
    for (N, k, B, kw) in ((64, 200, 5, {}), (64, 200, 100, {}), (48, 100, 33, {}), (33, 50, 7, {"maximize": True}),
                          (64, 120, 3, {"cutoff": 0.08}), (40, 64, 1, {})):
        costs = rng.random((B, N * N))
        want = plain.kbest(costs, N, N, k, **kw)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs[:4], N, N, k, **kw)
        assert (want[0][:4] == onf).all() and (want[1][:4] == or4c).all() and (bits(want[3][:4]) == bits(og)).all()
        for name, e in engs.items():
            got = e.kbest(costs, N, N, k, **kw)
            assert (got[0] == want[0]).all(), (name, N, k, B)
            for b in range(B):
                n = int(got[0][b])
                assert (got[1][b, :n] == want[1][b, :n]).all() and (bits(got[3][b, :n]) == bits(want[3][b, :n])).all(), (name, N, k, B, b)
                assert (got[2][b, :n] == want[2][b, :n]).all(), (name, N, k, B, b)  # square: every row has a real column
    for e in list(engs.values()) + [plain]:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 50, 40), (16, 16, 30, 700), (12, 12, 20, 64), (28, 10, 60, 33), (100, 100, 12, 6), (70, 20, 15, 5)])
def test_int8_tables_equal_the_int32_tables(shape):
    """KBEST_FLAG_TABLES_I8: the same row4col / col4row, one byte per entry, from every kernel (64-row, lane-per-child,
    small-problem, general-size), with unused slots filled with -1 -- plain and registered host buffers."""
    N, M, k, B = shape
    rng = np.random.default_rng(N * 1000 + M)
    costs = rng.uniform(0.0, 1.0, (B, N * M))
    if N >= 64:
        costs[0, :] = np.inf  # an infeasible problem: nf = 0, every slot is "unused"
        costs[0, : N * M : 7] = 1.0
    eng = pk.KBestEngine(0)
    nf, r4c, c4r, gain = eng.kbest(costs, N, M, k)
    nf8, r8, c8, gain8 = eng.kbest(costs, N, M, k, tables_i8=True)
    assert r8.dtype == np.int8 and c8.dtype == np.int8
    assert np.array_equal(nf, nf8) and np.array_equal(gain.view(np.int64), gain8.view(np.int64))
    assert np.array_equal(r4c, r8) and np.array_equal(c4r, c8)
    # registered buffers: the kernels write the byte tables straight into host memory
    import ctypes as C
    from probabilisticsemslam_amd import engine as E
    rr, cc = np.full((B, k, M), 99, np.int8), np.full((B, k, N), 99, np.int8)
    g2, n2 = np.zeros((B, k)), np.zeros(B, np.int32)
    eng.register_host(rr, cc, g2, n2)
    try:
        o = eng._opts(False, None, E.KBEST_FLAG_TABLES_I8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        cst = np.ascontiguousarray(costs)
        eng._check(eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(cst), None, k, p(rr), p(cc), p(g2), p(n2), None))
    finally:
        eng.unregister_host(rr, cc, g2, n2)
    assert np.array_equal(rr, r4c) and np.array_equal(cc, c4r) and np.array_equal(n2, nf)


@pytest.mark.gpu
def test_int8_tables_refused_beyond_127_rows():
    eng = pk.KBestEngine(0)
    costs = np.random.default_rng(3).uniform(0, 1, (2, 130 * 130))
    with pytest.raises(pk.KBestError, match="int8"):
        eng.kbest(costs, 130, 130, 4, tables_i8=True)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [1, 9, 40, 700])
def test_association_entry_on_registered_buffers(F):
    """kbest_assoc_probs_batch_f64 with the cost blocks and the probabilities in registered caller memory (read / written
    in place by the fused kernel) returns the same bits as with plain buffers -- which the other tests hold against the
    reference's goldens and the checker."""
    import ctypes as C
    k, nL, nM = 200, 20, 10
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM, seed=0xBEEF + F)
    nR = nL + nM
    raw = np.ascontiguousarray(np.concatenate(frames))
    h_nL, h_nM = np.full(F, nL, np.int32), np.full(F, nM, np.int32)
    h_coff = np.arange(F, dtype=np.int64) * nR * nM
    h_poff = np.arange(F, dtype=np.int64) * nM * (nL + 1)
    eng = pk.KBestEngine(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def call(hp, hnf):
        eng._check(eng.lib.kbest_assoc_probs_batch_f64(eng.ctx, F, p(h_nL), p(h_nM), p(raw), p(h_coff), k, p(hp), p(h_poff), p(hnf)))

    ref_p, ref_nf = np.zeros(F * nM * (nL + 1)), np.zeros(F, np.int32)
    call(ref_p, ref_nf)
    assert (ref_nf > 0).all() and np.allclose(ref_p.reshape(F, nM, nL + 1).sum(axis=2), 1.0, rtol=0, atol=1e-12)
    hp, hnf = np.full_like(ref_p, -7.0), np.zeros(F, np.int32)
    eng.register_host(raw, hp)
    try:
        call(hp, hnf)
        call(hp, hnf)
    finally:
        eng.unregister_host(raw, hp)
    assert np.array_equal(hp.view(np.int64), ref_p.view(np.int64)) and np.array_equal(hnf, ref_nf)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 120, 300), (32, 32, 200, 700), (16, 16, 50, 900), (12, 12, 30, 40), (40, 17, 60, 20),
                                   (100, 100, 25, 6), (90, 30, 40, 5)])
def test_column_order_of_the_enumeration_does_not_change_the_result(shape):
    """DESIGN section 2 point 8: the kernels enumerate in a column order of their own (dear columns first).  With
    KBEST_FLAG_NO_REORDER they walk the columns as the reference does.  Tie-free costs: both give the same tables, bit for bit
    (and the default is what every other test holds against the oracle)."""
    N, M, k, B = shape
    rng = np.random.default_rng(7 * N + M)
    costs = rng.uniform(0.0, 1.0, (B, N * M))
    eng = pk.KBestEngine(0)
    a = eng.kbest(costs, N, M, k)
    b = eng.kbest(costs, N, M, k, reorder=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[3].view(np.int64), b[3].view(np.int64))
    assert np.array_equal(a[1], b[1])
    real = b[2] < M  # col4row: rows on zero-padded columns may sit on another padded column
    assert np.array_equal(a[2] < M, real) and np.array_equal(a[2][real], b[2][real])
    # maximise + cutoff through the same switch
    a = eng.kbest(costs, N, M, k, maximize=True, cutoff=0.5)
    b = eng.kbest(costs, N, M, k, maximize=True, cutoff=0.5, reorder=False)
    n = a[0]
    assert np.array_equal(n, b[0])
    for i in range(B):
        assert np.array_equal(a[3][i, : n[i]].view(np.int64), b[3][i, : n[i]].view(np.int64)) and np.array_equal(a[1][i, : n[i]], b[1][i, : n[i]])
