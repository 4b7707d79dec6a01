"""GPU tests added in round 6: every line of bench.py's world > 1 path executed (two gloo ranks on GPU 0, weak and strong
scaling), the timed (relay) path pinned at full size against the checker, the narrow exchange of the multi-device entries
(int8 slices; subtree mode gains first), the gains-only merge on the device."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


# ------------------------------------------------------------------------------------ bench.py, world 2
@pytest.mark.parametrize("scaling,launcher", [("weak", "driver"), ("strong", "driver"), ("weak", "self")])
def test_bench_world2_over_gloo_on_one_gpu(scaling, launcher):
    """The deliverable's world > 1 branches -- rank-private slices of the seeded stream, the packed int8 slice per rank, ONE
    overlapped all-gather per step, the cross-rank checks (same global table on every rank; a neighbour's slice is what its
    matrices give), max-over-ranks timing, weak AND strong scaling -- with two ranks, both on GPU 0, the slices travelling through
    gloo (RCCL refuses two ranks on one GPU).  Launched exactly as the driver does (`python -m torch.distributed.run ...`), and once
    through bench.py's own launcher.  SURVEY 8(e)."""
    port = _free_port()
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", scaling]
    if launcher == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
        env = _clean_env(KBEST_BENCH_BACKEND="gloo")
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
        env = _clean_env(KBEST_BENCH_BACKEND="gloo", MASTER_PORT=str(port))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    per_rank = 1024 if scaling == "weak" else 512
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["matrices_per_gpu"] == per_rank
    assert out["parity_prune_vs_noprune"] is True
    c = out["collective"]
    assert c["backend"] == "gloo" and c["row4col_dtype"] == "int8"
    assert c["bytes_per_rank_per_step"] == per_rank * 200 * (8 + 64) + per_rank * 4
    assert c["bytes_inbound_per_rank_per_step"] == c["bytes_per_rank_per_step"]  # world - 1 = 1 slice comes in
    assert c["exposed_ms"] is not None and c["ms_per_step_without_the_gather"] > 0
    # whole-job value: both ranks' assignments over the slowest rank's time
    assert abs(out["value"] - 2 * per_rank * 200 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-9
    assert out["problems_per_s"] > 0 and out["roofline"]["frac"] > 0.05
    assert "cpu_baseline" not in out and "value_host_inclusive" not in out  # rank 0 at N = 1 only


# ------------------------------------------------------------------------------------ the timed path, pinned
def _dev_tables(eng, costs, N, M, k, tables_i8=False):
    import torch
    dev = torch.device("cuda", 0)
    B = costs.shape[0]
    tdt = torch.int8 if tables_i8 else torch.int32
    d_cost = torch.from_numpy(np.ascontiguousarray(costs)).to(dev)
    d_r = torch.full((B, k, M), -7, dtype=tdt, device=dev)
    d_c = torch.full((B, k, N), -7, dtype=tdt, device=dev)
    d_g = torch.full((B, k), float("nan"), dtype=torch.float64, device=dev)
    d_n = torch.full((B,), -7, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    eng.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=s.cuda_stream, tables_i8=tables_i8)
    torch.cuda.synchronize()
    return d_n.cpu().numpy(), d_r.cpu().numpy().astype(np.int32), d_c.cpu().numpy().astype(np.int32), d_g.cpu().numpy()


@pytest.mark.parametrize("name,i8", [("c4", False), ("c3", False), ("c4", True)])
def test_timed_relay_path_at_full_size_against_the_checker(name, i8):
    """What bench.py times -- kbest_batch_f64_dev on the FULL batch of C4 (1 024 x 64x64) / C3 (4 096 x 32x32), k = 200, which the
    launch plan runs as a relay (several workgroups per matrix in turn, the LDS handed on through HBM) -- against the checker on
    EVERY matrix: nf, row4col, col4row (zero-padded columns mapped to -1) and the gains' bits.  Also with int8 tables (what the
    multi-GPU step writes into its packed slice).  shortestPathCPP.cpp:571-644."""
    eng = pk.KBestEngine(0)
    costs, N, M, k = wl.dense_config(name)
    before = eng.relay_launches()
    nf, r4c, c4r, g = _dev_tables(eng, costs, N, M, k, tables_i8=i8)
    assert eng.relay_launches() == before + 1, "the launch was not a relay: this test no longer pins the timed path"
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (nf == onf).all()
    assert (r4c == or4c).all()
    assert (bits(g) == bits(og)).all()
    a, b = c4r.copy(), np.asarray(oc4r).copy()
    a[a >= M] = -1
    b[b >= M] = -1
    assert (a == b).all()
    eng.close()


# ------------------------------------------------------------------------------------ the multi-device entries' exchange
def test_multi_entry_exchanges_bytes_not_words(engine, monkeypatch):
    """Batch mode over logical devices: the slices that travel between the devices hold row4col as int8 (every index of a 64-row
    problem fits a byte): 8 + M bytes per solution instead of 8 + 4 M; KBEST_MULTI_WIDE=1 is round 5's int32 exchange.  Same
    tables either way, equal to the single-device entry's."""
    B, N, M, k = 70, 64, 64, 200
    costs = wl.dense_batch(B, N, M, 0x4D554D)
    want = engine.kbest(costs, N, M, k)
    sent = {}
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("KBEST_MULTI_WIDE", "1")
        multi = pk.KBestMulti([0, 0, 0, 0])
        got = multi.kbest(costs, N, M, k)
        assert multi.tables_agree()
        sent[wide], path = multi.exchange_bytes()
        assert path == 0
        multi.close()
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (bits(got[3]) == bits(want[3])).all()
        c = got[2].copy(); c[c >= M] = -1
        w = want[2].copy(); w[w >= M] = -1
        assert (c == w).all()
    monkeypatch.delenv("KBEST_MULTI_WIDE")
    pad = (B + 3) // 4
    assert sent[False] == 3 * (pad * k * 8 + pad * k * M + ((pad * 4 + 15) & ~15))  # three other devices' slices arrive
    assert sent[True] > 3.5 * sent[False]
    # rectangular / ragged problems (the path whose kernels write int32 tables: narrowed into the slice on the device)
    rng = np.random.default_rng(8)
    Br, Nr, Mr, kr = 41, 30, 12, 40
    nRow = rng.integers(12, Nr + 1, Br).astype(np.int32)
    nCol = np.minimum(rng.integers(1, Mr + 1, Br), nRow).astype(np.int32)
    packed = np.zeros((Br, Nr * Mr))
    for b in range(Br):
        packed[b, : nRow[b] * nCol[b]] = rng.random(int(nRow[b]) * int(nCol[b]))
    multi = pk.KBestMulti([0, 0, 0])
    nf, r4c, c4r, g = multi.kbest(packed, Nr, Mr, kr, nRow=nRow, nCol=nCol)
    assert multi.tables_agree()
    multi.close()
    for b in range(Br):
        n, m = int(nRow[b]), int(nCol[b])
        onf, or4c, _, og = ol.orc_kbest(packed[b, : n * m], n, m, kr)
        assert nf[b] == onf and (r4c[b, :onf, :m] == or4c[:onf]).all() and (bits(g[b, :onf]) == bits(og[:onf])).all()
        assert (r4c[b, onf:] == -1).all() and (r4c[b, :onf, m:] == -1).all()


@pytest.mark.parametrize("G,S", [(1, 4), (2, 2), (3, 8), (4, 4), (8, 8)])
def test_subtree_mode_gains_first(engine, monkeypatch, G, S):
    """Subtree mode, the north star's exchange: ONE all-gather of every shard's top-k COSTS (gain[k] + nf), the merge into the global
    k-best heap on every device, ONE sum all-reduce of the winners' rows -- against the whole-list exchange (KBEST_MULTI_WHOLE_LISTS),
    the single-device entry and the checker.  Bytes arriving at a device per matrix: 8 k S' (G - 1) + 2 k M (G - 1) / G instead of
    (8 + M) k S' (G - 1) (S' = shards per device)."""
    B, N, M, k = 7, 48, 48, 120
    costs = wl.dense_batch(B, N, M, 0x535543)
    costs[3, : N] = np.inf  # an infeasible matrix (a column of +inf): nf = 0 on every shard (cpp:588-593)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert onf[3] == 0
    res = {}
    for whole in (False, True):
        if whole:
            monkeypatch.setenv("KBEST_MULTI_WHOLE_LISTS", "1")
        multi = pk.KBestMulti([0] * G)
        got = multi.kbest(costs, N, M, k, subtree=True, n_shard=S)
        assert multi.tables_agree()
        sent, path = multi.exchange_bytes()
        multi.close()
        assert path == (2 if whole else 1)
        res[whole] = (got, sent)
        nf, r4c, c4r, g = got
        assert (nf == onf).all()
        for b in range(B):
            n = int(onf[b])
            assert (r4c[b, :n] == or4c[b, :n]).all() and (bits(g[b, :n]) == bits(og[b, :n])).all(), (whole, b)
            assert (r4c[b, n:] == -1).all() and (g[b, n:] == 0).all(), (whole, b)
    monkeypatch.delenv("KBEST_MULTI_WHOLE_LISTS")
    spd = (S + G - 1) // G
    up16 = lambda x: (x + 15) & ~15  # noqa: E731
    head = up16(spd * B * k * 8) + up16(spd * B * 4)
    assert res[False][1] == (G - 1) * head + 2 * B * k * M * (G - 1) // G
    assert res[True][1] == (G - 1) * (head + up16(spd * B * k * M))
    if G >= 3:
        assert res[True][1] > 1.5 * res[False][1]


def test_subtree_mode_exact_ties_take_the_whole_lists(engine):
    """Integer costs: candidates with exactly the same gain.  The gains-only merge cannot order them (the rule is (gain, row4col
    lexicographic)): the call notices -- every device merges the same gains -- and exchanges the whole lists, as round 5 did.  The
    multiset of gains and the validity of every assignment are the checker's."""
    rng = np.random.default_rng(17)
    B, N, M, k = 5, 10, 10, 64
    costs = np.floor(rng.random((B, N * M)) * 3)
    multi = pk.KBestMulti([0, 0])
    nf, r4c, c4r, g = multi.kbest(costs, N, M, k, subtree=True, n_shard=4)
    assert multi.tables_agree()
    sent, path = multi.exchange_bytes()
    multi.close()
    assert path == 2
    onf, or4c, _, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    for b in range(B):
        n = int(onf[b])
        assert nf[b] == n
        assert (np.sort(g[b, :n]) == np.sort(og[b, :n])).all()
        assert len({tuple(x) for x in r4c[b, :n].tolist()}) == n
        for s in range(n):
            assert len(set(r4c[b, s])) == M and costs[b].reshape(M, N)[np.arange(M), r4c[b, s]].sum() == g[b, s]


@pytest.mark.parametrize("S", [1, 2, 3, 8])
def test_device_merge_of_gains_is_the_global_kbest(engine, S):
    """kbest_merge_gains_f64_dev (C ABI; what one process per GPU calls between its all-gather of the costs and its all-reduce of
    the rows): every shard's call scatters its own winners; the SUM of the S byte tables is the single enumeration's table = the
    checker's; gains and counts come out whole from every call; integer costs set the tie word."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(60 + S)
    for (N, M, k, B, ties, maximize) in ((16, 16, 50, 6, False, False), (40, 12, 120, 3, False, False), (10, 10, 64, 5, True, False),
                                          (24, 24, 30, 4, False, True), (5, 5, 200, 3, False, False)):
        costs = rng.random((B, N * M))
        if ties:
            costs = np.floor(costs * 3)
        lists = [engine.kbest(costs, N, M, k, root_shard=(s, S), maximize=maximize, tables_i8=True) for s in range(S)]
        tg = torch.from_numpy(np.stack([l[3] for l in lists])).to(dev)
        tn = torch.from_numpy(np.stack([l[0] for l in lists])).to(dev)
        total = torch.zeros((B, k, M), dtype=torch.int32, device=dev)
        first = None
        any_tied = 0
        for s in range(S):
            own = torch.from_numpy(np.ascontiguousarray(lists[s][1])).to(dev)
            og = torch.full((B, k), float("nan"), dtype=torch.float64, device=dev)
            orow = torch.zeros((B, k, M), dtype=torch.int8, device=dev)
            on = torch.full((B,), -7, dtype=torch.int32, device=dev)
            tied = torch.zeros(4, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            engine.merge_gains_dev(B, S, k, M, tg, tn, s, own, og, orow, on, tied, maximize=maximize)
            torch.cuda.synchronize()
            total += orow.to(torch.int32)
            any_tied |= int(tied[0])
            if first is None:
                first = (og.cpu().numpy(), on.cpu().numpy())
            else:  # gains and counts: the same from every shard's call
                n0 = first[1]
                assert (on.cpu().numpy() == n0).all()
                for b in range(B):
                    assert (bits(og.cpu().numpy()[b, : n0[b]]) == bits(first[0][b, : n0[b]])).all()
        onf, or4c, _, ogain, _ = ol.orc_kbest_batch(costs, N, M, k, maximize=maximize)
        assert (first[1] == onf).all()
        if ties:
            assert any_tied == 1  # (ties inside one shard's list are reported as well)
            for b in range(B):
                n = int(onf[b])
                assert (np.sort(first[0][b, :n]) == np.sort(ogain[b, :n])).all()
            continue
        assert any_tied == 0
        tot = total.cpu().numpy()
        for b in range(B):
            n = int(onf[b])
            assert (bits(first[0][b, :n]) == bits(ogain[b, :n])).all() and (tot[b, :n] == or4c[b, :n]).all()
            assert (tot[b, n:] == 0).all()


def test_engine_in_a_process_group_world2_narrow_exchange(engine, tmp_path):
    """The world-2 gloo group of round 5 with the round-6 exchange: batch mode with int8 slices, subtree mode gains first (the
    engine's int8 tables go in as they are) -- every rank ends up with the single-rank engine's tables = the checker's."""
    B, Bsub, world = 13, 3, 2
    out = str(tmp_path / "dist6")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = _clean_env(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                         KBEST_DIST_NARROW="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_engine_worker.py"), out, str(B), str(Bsub)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    cases = {"batch": wl.dense_config("c2", B=B), "sub": wl.dense_config("c2", B=Bsub),
             "sub64": (wl.dense_config("c4", B=2)[0], 64, 64, 60)}
    for tag, (costs, N, M, k) in cases.items():
        nf, r4c, c4r, g = engine.kbest(costs, N, M, k)
        for rank in range(world):
            z = np.load(f"{out}.rank{rank}.npz")
            assert (z[f"{tag}_nf"] == nf).all(), (tag, rank)
            assert (z[f"{tag}_r"] == r4c).all(), (tag, rank)
            assert (bits(z[f"{tag}_g"]) == bits(g)).all(), (tag, rank)
            if tag != "batch":
                assert str(z[f"{tag}_path"]) == "gains_first", (tag, rank)


# ------------------------------------------------------------------------------------ exact ties beyond 64, k at a kernel's limit
def engine_with(monkeypatch, **env):
    for key, val in env.items():
        monkeypatch.setenv(key, str(val))
    eng = pk.KBestEngine(0)
    for key in env:
        monkeypatch.delenv(key)
    return eng


TIE_ROUTES = [{}, {"KBEST_NO_LANE": 1, "KBEST_NO_SMALL": 1}, {"KBEST_FORCE_SMALL": 1}, {"KBEST_FORCE_LANE": 1}, {"KBEST_FORCE_WIDE": 1}]


@pytest.mark.parametrize("shape", [(12, 12, 60, 8), (8, 8, 20, 4), (9, 9, 40, 5), (12, 6, 25, 4), (10, 8, 30, 5)])
def test_tie_levels_of_more_than_64_members_are_completed_on_every_route(monkeypatch, shape):
    """Round 6: a synchronous entry completes a gain level that straddles slot k in steps -- k + 64, k + 256, k + 1 024 solutions --
    so levels of hundreds of members (integer costs from a small range) now come back as the ONE answer on every route: the
    checker's canonical k best, bit for bit; levels beyond the cap (or of more than 1 024 members) stay flagged.  The cases must
    contain levels that round 5's single step of 64 left open.  shortestPathCPP.cpp:30-42, 574."""
    N, M, k, hi = shape
    rng = np.random.default_rng(31 * N + k)
    B = 10
    costs = rng.integers(0, hi, size=(B, N * M)).astype(np.float64)
    want = [ol.canonical_kbest(costs[b], N, M, k) for b in range(B)]
    want64 = [ol.canonical_kbest(costs[b], N, M, k, cap=64) for b in range(B)]
    newly = sum(1 for w, w64 in zip(want, want64) if w[4] and not w64[4])
    assert newly > 0, "no level between 64 and 1 024 members beyond k in this case"
    first = None
    for knobs in TIE_ROUTES:
        eng = engine_with(monkeypatch, **knobs)
        nf, r4c, c4r, g, fl = eng.kbest(costs, N, M, k, tie_flags=True, canonical_ties=True)
        for b in range(B):
            wn, wr, wg, boundary, resolved = want[b]
            assert nf[b] == wn and (bits(g[b, :wn]) == bits(wg)).all(), (knobs, b)
            assert bool(fl[b] & pk.engine.KBEST_TIE_BOUNDARY) == boundary, (knobs, b, fl[b])
            if boundary and not resolved:
                assert fl[b] & pk.engine.KBEST_TIE_UNRESOLVED, (knobs, b)
                continue
            assert not (fl[b] & pk.engine.KBEST_TIE_UNRESOLVED) and bool(fl[b] & pk.engine.KBEST_TIE_RESOLVED) == boundary, (knobs, b, fl[b])
            assert (r4c[b, :wn] == wr).all(), (knobs, b)
        if first is None:
            first = (r4c.copy(), g.copy(), fl.copy())
        else:
            ok = (first[2] & pk.engine.KBEST_TIE_UNRESOLVED) == 0
            assert (first[0][ok] == r4c[ok]).all() and (bits(first[1][ok]) == bits(g[ok])).all() and (first[2][ok] == fl[ok]).all(), knobs


def test_resolve_ties_dev_completes_the_device_tables(engine):
    """kbest_resolve_ties_dev with KBEST_FLAG_CANONICAL_TIES (the engine's own rule; the default -- the reference's answer -- is
    test_reference_ties_behind_the_device_entry_and_on_two_devices): the second call behind the asynchronous entry.  Integer costs through kbest_batch_f64_dev leave
    KBEST_TIE_BOUNDARY flags; the helper completes those levels in the DEVICE tables: afterwards they equal the synchronous
    entry's (= the checker's canonical k best) and the flags say RESOLVED / UNRESOLVED.  Also through a multi-device batch call."""
    import torch
    dev = torch.device("cuda", 0)
    E = pk.engine
    rng = np.random.default_rng(99)
    for (N, M, k, hi, B, i8) in ((8, 8, 20, 4, 40, False), (16, 16, 50, 30, 300, False), (40, 40, 60, 200, 24, True), (12, 7, 30, 25, 30, False)):
        costs = rng.integers(0, hi, size=(B, N * M)).astype(np.float64)
        want = engine.kbest(costs, N, M, k, tie_flags=True, tables_i8=i8, canonical_ties=True)
        tdt = torch.int8 if i8 else torch.int32
        d_cost = torch.from_numpy(costs).to(dev)
        d_r = torch.empty((B, k, M), dtype=tdt, device=dev)
        d_c = torch.empty((B, k, N), dtype=tdt, device=dev)
        d_g = torch.empty((B, k), dtype=torch.float64, device=dev)
        d_n = torch.empty(B, dtype=torch.int32, device=dev)
        d_f = torch.zeros(B, dtype=torch.int32, device=dev)
        s = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        engine.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=s.cuda_stream, d_tie_flags=d_f, tables_i8=i8)
        torch.cuda.synchronize()
        f0 = d_f.cpu().numpy()
        assert ((f0 & E.KBEST_TIE_BOUNDARY) != 0).sum() > 0 and ((f0 & (E.KBEST_TIE_RESOLVED | E.KBEST_TIE_UNRESOLVED)) == 0).all()
        engine.resolve_ties_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_f, stream=s.cuda_stream, tables_i8=i8, canonical_ties=True)
        torch.cuda.synchronize()
        f1 = d_f.cpu().numpy()
        assert (f1 == want[4]).all()
        ok = (f1 & E.KBEST_TIE_UNRESOLVED) == 0
        assert ok.sum() > 0
        assert (d_r.cpu().numpy()[ok] == want[1][ok]).all() and (bits(d_g.cpu().numpy()[ok]) == bits(want[3][ok])).all()
        cg, cw = d_c.cpu().numpy()[ok].astype(np.int32), want[2][ok].astype(np.int32)
        cg[cg >= M] = -1
        cw[cw >= M] = -1
        assert (cg == cw).all()
    # a RAGGED batch (per-problem shapes, packed blocks) through the same helper
    B, maxRow, maxCol, k = 60, 12, 9, 25
    nRow = rng.integers(4, maxRow + 1, B).astype(np.int32)
    nCol = np.array([int(rng.integers(2, min(r, maxCol) + 1)) for r in nRow], np.int32)
    blocks = [rng.integers(0, 4, int(r) * int(c)).astype(np.float64) for r, c in zip(nRow, nCol)]
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum([len(b_) for b_ in blocks[:-1]])
    flat = np.concatenate(blocks)
    want = engine.kbest(flat, maxRow, maxCol, k, nRow=nRow, nCol=nCol, costOff=off, tie_flags=True, canonical_ties=True)
    d_cost = torch.from_numpy(flat).to(dev)
    d_nR, d_nC, d_off = torch.from_numpy(nRow).to(dev), torch.from_numpy(nCol).to(dev), torch.from_numpy(off).to(dev)
    d_r = torch.full((B, k, maxCol), -1, dtype=torch.int32, device=dev)
    d_c = torch.full((B, k, maxRow), -1, dtype=torch.int32, device=dev)
    d_g = torch.zeros((B, k), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    engine.kbest_dev(d_cost, B, maxRow, maxCol, k, d_r, d_c, d_g, d_n, stream=st, d_tie_flags=d_f, d_nRow=d_nR, d_nCol=d_nC, d_costOff=d_off)
    engine.resolve_ties_dev(d_cost, B, maxRow, maxCol, k, d_r, d_c, d_g, d_f, stream=st, d_nRow=d_nR, d_nCol=d_nC, d_costOff=d_off, canonical_ties=True)
    torch.cuda.synchronize()
    f1, nfd = d_f.cpu().numpy(), d_n.cpu().numpy()
    assert (f1 == want[4]).all() and ((f1 & E.KBEST_TIE_RESOLVED) != 0).sum() > 0 and (nfd == want[0]).all()
    rg, gg = d_r.cpu().numpy(), d_g.cpu().numpy()
    for b in range(B):
        if f1[b] & E.KBEST_TIE_UNRESOLVED:
            continue
        n, m = int(nfd[b]), int(nCol[b])
        assert (rg[b, :n, :m] == want[1][b, :n, :m]).all() and (bits(gg[b, :n]) == bits(want[3][b, :n])).all(), b
    # the multi-device batch entry completes tied levels by itself: in the caller's tables AND in the devices' slices
    N, M, k, hi, B = 10, 10, 30, 4, 50
    costs = rng.integers(0, hi, size=(B, N * M)).astype(np.float64)
    want = engine.kbest(costs, N, M, k, tie_flags=True, canonical_ties=True)
    multi = pk.KBestMulti([0, 0, 0])
    got = multi.kbest(costs, N, M, k, canonical_ties=True)
    assert multi.tables_agree()
    fl = multi.last_tie_flags()
    multi.close()
    assert (fl == want[4]).all() and ((fl & E.KBEST_TIE_RESOLVED) != 0).sum() > 0
    ok = (fl & E.KBEST_TIE_UNRESOLVED) == 0
    assert (got[0] == want[0]).all() and (got[1][ok] == want[1][ok]).all() and (bits(got[3][ok]) == bits(want[3][ok])).all()


def _route(eng, costs, N, M, k, sync=False):
    """(nf, row4col, col4row, gain, flags) and the route: through the asynchronous entry (sync=False) or kbest_batch_f64."""
    if sync:
        out = eng.kbest(costs, N, M, k, tie_flags=True)
        return out, eng.last_route()
    import torch
    dev = torch.device("cuda", 0)
    B = costs.shape[0]
    d_cost = torch.from_numpy(np.ascontiguousarray(costs)).to(dev)
    d_r = torch.full((B, k, M), -1, dtype=torch.int32, device=dev)
    d_c = torch.full((B, k, N), -1, dtype=torch.int32, device=dev)
    d_g = torch.zeros((B, k), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=torch.cuda.current_stream().cuda_stream, d_tie_flags=d_f)
    torch.cuda.synchronize()
    return (d_n.cpu().numpy(), d_r.cpu().numpy(), d_c.cpu().numpy(), d_g.cpu().numpy(), d_f.cpu().numpy()), eng.last_route()


def test_k_at_a_kernel_limit_keeps_its_kernel(monkeypatch):
    """with exact ties checked a launch enumerates k + 1 solutions, which moved every k limit down by one.  Now, through
    the ASYNCHRONOUS entry, the caller's k decides: at the largest k a kernel takes the launch runs on THAT kernel without the extra
    solution and its problems carry KBEST_TIE_UNCHECKED; one below, the extra solution is enumerated.  The SYNCHRONOUS entry at
    that k takes a kernel that fits k + 1 (the tie at slot k stays checked).  Results: the checker's, bit for bit."""
    E = pk.engine
    rng = np.random.default_rng(4)
    cases = [("fast", E.KBEST_ROUTE_FAST, dict(KBEST_NO_LANE=1, KBEST_NO_SMALL=1), 40, 40, 2),
             ("small", E.KBEST_ROUTE_SMALL, dict(), 30, 10, 3),
             ("lane", E.KBEST_ROUTE_LANE, dict(KBEST_FORCE_LANE=1), 12, 12, 5)]
    for name, bit, knobs, N, M, B in cases:
        eng = engine_with(monkeypatch, **knobs)
        costs = rng.random((B, N * M))
        # the largest k this kernel takes for the shape: bisection on the route (monotone in k)
        lo, hi = 8, 8192
        assert _route(eng, costs, N, M, lo)[1] & bit, name
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if _route(eng, costs, N, M, mid)[1] & bit:
                lo = mid
            else:
                hi = mid
        kmax = lo
        assert 64 <= kmax < 8192, (name, kmax)
        (nf, r4c, c4r, g, fl), route = _route(eng, costs, N, M, kmax)
        assert route & bit and not (route & E.KBEST_ROUTE_EXTRA), (name, kmax, route)
        assert (fl & E.KBEST_TIE_UNCHECKED).all(), (name, kmax)
        onf, or4c, _, og, _ = ol.orc_kbest_batch(costs, N, M, kmax)
        assert (nf == onf).all()
        for b in range(B):
            n = int(onf[b])
            assert (r4c[b, :n] == or4c[b, :n]).all() and (bits(g[b, :n]) == bits(og[b, :n])).all(), (name, b)
        (nf2, r2, c2, g2, fl2), route2 = _route(eng, costs, N, M, kmax - 1)
        assert route2 & bit and route2 & E.KBEST_ROUTE_EXTRA and not (fl2 & E.KBEST_TIE_UNCHECKED).any(), (name, kmax - 1, route2)
        (nf3, r3, c3, g3, fl3), route3 = _route(eng, costs, N, M, kmax + 1)   # beyond: another kernel, still the checker's answer
        assert not (route3 & bit)
        n3 = np.minimum(nf, nf3)
        for b in range(B):
            assert (r3[b, : n3[b]] == r4c[b, : n3[b]]).all() and (bits(g3[b, : n3[b]]) == bits(g[b, : n3[b]])).all()
        # the synchronous entry at kmax: checked (the extra solution is enumerated -- by whichever kernel takes kmax + 1)
        (nf4, r4, c4, g4, fl4), route4 = _route(eng, costs, N, M, kmax, sync=True)
        assert route4 & E.KBEST_ROUTE_EXTRA and not (fl4 & E.KBEST_TIE_UNCHECKED).any(), (name, route4)
        assert (nf4 == nf).all() and (r4 == r4c).all() and (bits(g4) == bits(g)).all()


def test_fused_association_entry_takes_k_1024():
    """kbest_assoc_probs_batch_f64_dev promises k <= 1 024 (kbest_c.h); round 5 refused k = 1 024 because it enumerated k + 1.
    KITTI-like frames at k = 1 024 and 1 023 through the fused enumeration kernel (bounded walk off) and through the default
    route: the checker's probabilities (assignment.cpp:547-683), rel 1e-12."""
    import torch
    dev = torch.device("cuda", 0)
    F, nL, nM = 24, 20, 10
    nR = nL + nM
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM)
    raw = np.ascontiguousarray(np.concatenate(frames))
    d_cost = torch.from_numpy(raw).to(dev)
    d_nL = torch.full((F,), nL, dtype=torch.int32, device=dev)
    d_nM = torch.full((F,), nM, dtype=torch.int32, device=dev)
    d_nRow = torch.full((F,), nR, dtype=torch.int32, device=dev)
    d_coff = torch.arange(F, dtype=torch.int64, device=dev) * (nR * nM)
    d_poff = torch.arange(F, dtype=torch.int64, device=dev) * (nM * (nL + 1))
    for k in (1024, 1023):
        want = []
        for f in frames:
            cond, idx = ol.condition_costs(f, nR, nM)
            cl = len(idx) - nM
            q, _ = ol.assignment_prob(cond, cl, nM, k)
            full = np.zeros((nM, nL + 1))
            full[:, np.asarray(idx[:cl], dtype=np.int64)] = q[:, :cl]
            full[:, nL] = q[:, cl]
            want.append(full)
        for knobs in ({"KBEST_NO_BNB": "1"}, {}):
            for key, val in knobs.items():
                os.environ[key] = val
            eng = pk.KBestEngine(0)
            for key in knobs:
                del os.environ[key]
            d_probs = torch.zeros(F * nM * (nL + 1), dtype=torch.float64, device=dev)
            d_nf = torch.zeros(F, dtype=torch.int32, device=dev)
            d_fl = torch.zeros(F, dtype=torch.int32, device=dev)
            eng.reserve_assoc(F, nR, nM, k)
            eng.set_assoc_tie_flags_dev(d_fl)
            torch.cuda.synchronize()
            eng.assoc_probs_dev(F, nR, nM, d_nL, d_nM, d_nRow, d_cost, d_coff, k, d_probs, d_poff, d_nf, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert (d_nf.cpu().numpy() >= 0).all(), (k, knobs)
            p = d_probs.cpu().numpy().reshape(F, nM, nL + 1)
            for i in range(F):
                np.testing.assert_allclose(p[i], want[i], rtol=1e-12, atol=1e-300, err_msg=str((k, knobs, i)))
            if knobs and k == 1024:  # the fused enumeration kernel at its limit: no solution behind the k-th
                assert (d_fl.cpu().numpy() & pk.engine.KBEST_TIE_UNCHECKED).all()
            eng.close()


# ------------------------------------------------------------------------------------ the reference's own order (kbest_exact.hip)
def test_reference_order_reproduces_the_compiled_reference_goldens(engine, golden):
    """KBEST_FLAG_REFERENCE_ORDER: the reference's algorithm as it stands (padded N x N problem, one heap of fully solved
    hypotheses with libstdc++'s sift rules).  Every case of kbest_golden.npz -- the outputs of the UNMODIFIED reference -- slot for
    slot and bit for bit, col4row included WITHOUT mapping the padded columns (SURVEY quirk 6 does not apply to this kernel)."""
    E = pk.engine
    for name in golden.names:
        c = golden.case(name)
        cost = np.array(c["cost"], dtype=np.float64)
        if not np.isfinite(cost).all() and np.isnan(cost).any():
            continue
        nf, r4c, c4r, g = engine.kbest(cost.reshape(1, -1), c["N"], c["M"], c["k"], maximize=c["maximize"], cutoff=c["cutoff"],
                                      reference_order=True)
        assert engine.last_route() == E.KBEST_ROUTE_EXACT, name
        n = c["nf"]
        assert nf[0] == n, (name, nf[0], n)
        assert (r4c[0, :n] == c["row4col"]).all(), name
        assert (c4r[0, :n] == c["col4row"]).all(), name
        assert (bits(g[0, :n]) == bits(c["gain"])).all(), name


@pytest.mark.parametrize("shape", [(6, 6, 40, 3), (9, 9, 60, 4), (12, 7, 50, 5), (20, 20, 120, 30), (30, 10, 200, 12)])
def test_reference_order_on_exact_ties_is_the_heap_order(engine, shape):
    """Integer costs: masses of exactly equal gains.  With KBEST_FLAG_REFERENCE_ORDER the tables are the checker's -- whose pop order
    of equal gains is pinned to the compiled reference's std::priority_queue (tests/test_oracle_golden.py) -- slot for slot: the
    assignments IN THE REFERENCE'S ORDER, the raw col4row, the gains' bits, the push count; also with a cutoff and maximising.
    The default rule ((gain, row4col) lexicographic) gives the same gains and, as a rule, another order: both are checked.
    shortestPathCPP.cpp:30-42, 574."""
    N, M, k, hi = shape
    rng = np.random.default_rng(7 * N + k)
    B = 9
    costs = rng.integers(0, hi, size=(B, N * M)).astype(np.float64)
    differs = 0
    for kw in ({}, {"cutoff": float(hi)}, {"maximize": True}):
        nf, r4c, c4r, g, pushed = engine.kbest(costs, N, M, k, reference_order=True, count_pushed=True, **kw)
        onf, or4c, oc4r, og, opushed = ol.orc_kbest_batch(costs, N, M, k, **kw)
        assert (nf == onf).all() and (pushed == opushed).all(), kw
        for b in range(B):
            n = int(onf[b])
            assert (r4c[b, :n] == or4c[b, :n]).all(), (kw, b)
            assert (c4r[b, :n] == oc4r[b, :n]).all(), (kw, b)
            assert (bits(g[b, :n]) == bits(og[b, :n])).all(), (kw, b)
        d = engine.kbest(costs, N, M, k, canonical_ties=True, **kw)   # the engine's own rule: same gains, its own order of ties
        assert (d[0] == onf).all()
        for b in range(B):
            n = int(onf[b])
            assert (bits(d[3][b, :n]) == bits(og[b, :n])).all(), (kw, b)
            differs += int((d[1][b, :n] != or4c[b, :n]).any())
    assert differs > 0  # (the two orders of ties are different rules)


def test_host_tie_flags_with_modes_that_check_no_ties(engine):
    """kbest_batch_f64's tie_flags is a HOST array.  In the modes without a tie check (the reference's own order, push counting) it comes
    back zeroed -- it used to travel down to the launch as if it were a device pointer (hipMemsetAsync: invalid argument)."""
    rng = np.random.default_rng(2)
    C_ = rng.integers(0, 5, (3, 36)).astype(np.float64)
    for kw in ({"reference_order": True}, {"count_pushed": True}, {"tie_check": False}):
        out = engine.kbest(C_, 6, 6, 30, tie_flags=True, **kw)
        assert (out[-1] == 0).all(), kw
        assert (out[0] == 30).all(), kw


def test_reference_order_through_the_device_entry_and_ragged(engine):
    """The same kernel behind kbest_batch_f64_dev (kbest_reserve_exact first), on a ragged batch with an infeasible problem, a
    problem with fewer than k assignments and int8 tables."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    B, maxRow, maxCol, k = 40, 14, 9, 60
    nRow = rng.integers(2, maxRow + 1, B).astype(np.int32)
    nCol = np.array([int(rng.integers(1, min(r, maxCol) + 1)) for r in nRow], np.int32)
    nRow[3], nCol[3] = 3, 3
    blocks = [rng.integers(0, 6, int(r) * int(c)).astype(np.float64) for r, c in zip(nRow, nCol)]
    blocks[7][: int(nRow[7])] = np.inf
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum([len(b_) for b_ in blocks[:-1]])
    d_cost = torch.from_numpy(np.concatenate(blocks)).to(dev)
    d_nR, d_nC, d_off = torch.from_numpy(nRow).to(dev), torch.from_numpy(nCol).to(dev), torch.from_numpy(off).to(dev)
    d_r = torch.full((B, k, maxCol), -7, dtype=torch.int8, device=dev)
    d_c = torch.full((B, k, maxRow), -7, dtype=torch.int8, device=dev)
    d_g = torch.zeros((B, k), dtype=torch.float64, device=dev)
    d_n = torch.full((B,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    engine.kbest_dev(d_cost, B, maxRow, maxCol, k, d_r, d_c, d_g, d_n, stream=torch.cuda.current_stream().cuda_stream,
                     d_nRow=d_nR, d_nCol=d_nC, d_costOff=d_off, tables_i8=True, reference_order=True)
    torch.cuda.synchronize()
    assert engine.last_route() == pk.engine.KBEST_ROUTE_EXACT
    nf, r4c, c4r, g = d_n.cpu().numpy(), d_r.cpu().numpy(), d_c.cpu().numpy(), d_g.cpu().numpy()
    for b in range(B):
        n_, m_ = int(nRow[b]), int(nCol[b])
        wn, wr, wc, wg = ol.orc_kbest(blocks[b], n_, m_, k)
        assert nf[b] == wn, b
        assert (r4c[b, :wn, :m_] == wr[:wn]).all() and (c4r[b, :wn, :n_] == wc[:wn]).all() and (bits(g[b, :wn]) == bits(wg[:wn])).all(), b
    assert nf[7] == 0 and nf[3] == 6


def test_reference_order_large_host_batch(engine):
    """A batch that the host entry would send through the GPU in pieces with narrow staging (1 024 x 20x20, k = 200: 32 MB of tables)
    runs as ONE launch of the reference-order kernel when that order is asked for; also through the multi-device entry."""
    rng = np.random.default_rng(8)
    B, N, M, k = 1024, 20, 20, 200
    costs = rng.integers(0, 40, size=(B, N * M)).astype(np.float64)
    nf, r4c, c4r, g = engine.kbest(costs, N, M, k, reference_order=True)
    assert engine.last_route() == pk.engine.KBEST_ROUTE_EXACT
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    assert (nf == onf).all() and (r4c == or4c).all() and (c4r == oc4r).all() and (bits(g) == bits(og)).all()
    import ctypes as C
    multi = pk.KBestMulti([0, 0])
    o = pk.engine.KBestOpts()
    multi.lib.kbest_default_opts(C.byref(o))
    o.flags = pk.engine.KBEST_FLAG_REFERENCE_ORDER
    Bm = 64
    r2, c2, g2, n2 = np.zeros((Bm, k, M), np.int32), np.zeros((Bm, k, N), np.int32), np.zeros((Bm, k)), np.zeros(Bm, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    cst = np.ascontiguousarray(costs[:Bm])
    rc = multi.lib.kbest_batch_f64_multi(multi.m, C.byref(o), Bm, N, M, None, None, p(cst), k, p(r2), p(c2), p(g2), p(n2))
    assert rc == 0, multi.lib.kbest_multi_last_error(multi.m)
    assert multi.tables_agree()
    multi.close()
    assert (n2 == onf[:Bm]).all() and (r2 == or4c[:Bm]).all() and (c2 == oc4r[:Bm]).all() and (bits(g2) == bits(og[:Bm])).all()


def test_more_than_1024_rows_run_on_the_reference_order_kernel(engine):
    """kBest2D has no size limit in the reference (cpp:571-644).  Beyond KBEST_MAX_DIM_WIDE (1 024 rows) the reference-order kernel
    takes the problem: 1 100 x 6 and 1 300 x 3, k = 8, against the checker, slot for slot (raw col4row included)."""
    rng = np.random.default_rng(5)
    for (N, M, k, B) in ((1100, 6, 8, 2), (1300, 3, 5, 1)):
        costs = rng.random((B, N * M))
        nf, r4c, c4r, g = engine.kbest(costs, N, M, k)
        assert engine.last_route() == pk.engine.KBEST_ROUTE_EXACT
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
        assert (nf == onf).all() and (r4c == or4c).all() and (c4r == oc4r).all() and (bits(g) == bits(og)).all(), (N, M)


def test_shims_in_reference_order(tmp_path):
    """KBEST_SHIM_REFERENCE_ORDER=1: the reference-named C++ shims (kBest2D / kBest2DCutoff of include/kbest_shims.hpp) answer through
    the reference-order kernel.  The compiled C++ caller of tests/cpp/shim_drop_in.cpp prints the same lines with and without it on
    tie-free costs (8x8 and a rectangular 9x5 problem, whose col4row names padded columns: the shim maps nothing)."""
    exe = str(tmp_path / "shim_drop_in")
    libdir = os.path.join(ROOT, "probabilisticsemslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_drop_in.cpp"),
                           "-o", exe, "-L", libdir, "-l:libkbest_amd.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64"])
    for args in (["8", "8", "10", "12345"], ["9", "5", "12", "777"]):
        a = subprocess.check_output([exe] + args, text=True, env=_clean_env()).splitlines()
        b = subprocess.check_output([exe] + args, text=True, env=_clean_env(KBEST_SHIM_REFERENCE_ORDER="1")).splitlines()
        nf = int(a[0].split()[2])
        assert a[0] == b[0] and nf > 0
        for s in range(nf):  # gains and row4col identical; col4row identical on the real columns (padded columns: the reference's names in b)
            ta, tb = a[1 + s].split(), b[1 + s].split()
            i_c4r = ta.index("c4r")
            assert ta[: i_c4r] == tb[: i_c4r], (args, s)
            M = int(args[1])
            ca = [int(x) if int(x) < M else -1 for x in ta[i_c4r + 1:]]
            cb = [int(x) if int(x) < M else -1 for x in tb[i_c4r + 1:]]
            assert ca == cb, (args, s)


def test_association_probabilities_in_reference_order():
    """kbest_set_reference_order: assignmentProb / getAssignmentProbs with the k best enumerated by the reference-order kernel.  On
    integer-cost frames -- exact ties across slot k in most of them -- the probabilities are the checker's own (whose kBest2DCutoff
    pops equal gains in the compiled reference's heap order), rel 1e-12: the reference's answer, not the engine's rule.
    assignment.cpp:547-683."""
    eng = pk.KBestEngine(0)
    eng.set_reference_order(True)
    rng = np.random.default_rng(44)
    checked = ties = 0
    for (nL, nM, k, hi) in ((6, 3, 20, 6), (12, 5, 100, 8), (20, 10, 200, 12), (20, 10, 50, 4)):
        nR = nL + nM
        frames = []
        for _ in range(40):
            C_ = np.full(nR * nM, np.inf)
            for c in range(nM):
                near = rng.random(nL) < 4.0 / nL
                near[c % nL] = True
                C_[c * nR: c * nR + nL] = np.where(near, rng.integers(0, hi, nL), 60 + rng.integers(0, 400, nL)).astype(np.float64)
                C_[c * nR + nL + c] = 10.0
            frames.append(C_)
        P, nf = eng.weights(frames, [nL] * len(frames), [nM] * len(frames), k, condition=True)
        for f, fr in enumerate(frames):
            cond, idx = ol.condition_costs(fr, nR, nM)
            cl = len(idx) - nM
            q, n = ol.assignment_prob(cond, cl, nM, k)
            full = np.zeros((nM, nL + 1))
            full[:, np.asarray(idx[:cl], dtype=np.int64)] = q[:, :cl]
            full[:, nL] = q[:, cl]
            assert nf[f] == n, (nL, nM, k, f)
            np.testing.assert_allclose(P[f], full, rtol=1e-12, atol=1e-300, err_msg=str((nL, nM, k, f)))
            checked += 1
            ties += int(ol.canonical_kbest(cond, cl + nM, nM, k, cutoff=42.0)[3])
    assert checked == 160 and ties > 20  # (the generator does put exact ties across slot k)
    eng.close()


def _named(c4r, M):
    """col4row with the padded columns' names dropped (which padded column a left-over row sits on is immaterial)."""
    return np.where(c4r >= M, -1, c4r)


@pytest.mark.parametrize("shape", [(6, 6, 40, 3), (12, 7, 50, 5), (20, 20, 120, 30), (30, 10, 200, 12), (64, 64, 100, 9)])
def test_reference_ties_is_the_references_answer_on_every_problem(engine, shape):
    """KBEST_FLAG_REFERENCE_TIES: the batch on the fast kernels, then every problem with an exact tie among its k + 1 best gains again
    on the reference-order kernel.  A batch of integer-cost problems (ties) and continuous ones (none), alternating: EVERY problem's
    tables are the checker's slot for slot -- gains' bits, row4col in the reference's heap order, col4row (raw on the re-run problems)
    --, the tied problems are flagged KBEST_TIE_REFERENCE and the continuous ones are not touched.  Plain, with a cutoff, maximising."""
    E = pk.engine
    N, M, k, hi = shape
    rng = np.random.default_rng(11 * N + k)
    B = 12
    costs = rng.random((B, N * M)) * hi
    costs[::2] = rng.integers(0, hi, size=(B // 2, N * M)).astype(np.float64)
    for kw in ({}, {"cutoff": float(hi)}, {"maximize": True}):
        nf, r4c, c4r, g, fl = engine.kbest(costs, N, M, k, reference_ties=True, tie_flags=True, **kw)
        onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k, **kw)
        assert (nf == onf).all(), kw
        for b in range(B):
            n = int(onf[b])
            assert (bits(g[b, :n]) == bits(og[b, :n])).all(), (kw, b)
            assert (r4c[b, :n] == or4c[b, :n]).all(), (kw, b, hex(int(fl[b])))
            assert (_named(c4r[b, :n], M) == _named(oc4r[b, :n], M)).all(), (kw, b)
            tied_inside = n > 1 and bool((og[b, 1:n] == og[b, : n - 1]).any())
            if fl[b] & E.KBEST_TIE_REFERENCE:
                assert (c4r[b, :n] == oc4r[b, :n]).all(), (kw, b)
                assert not (fl[b] & (E.KBEST_TIE_BOUNDARY | E.KBEST_TIE_UNRESOLVED | E.KBEST_TIE_UNORDERED))
            else:
                assert fl[b] == 0 and not tied_inside, (kw, b, hex(int(fl[b])))
        assert (fl[1::2] == 0).all(), kw                       # continuous costs: nothing ties, nothing is run again
        assert (fl[::2] & E.KBEST_TIE_REFERENCE).any(), kw     # integer costs: ties


def test_reference_ties_behind_the_device_entry_and_on_two_devices(engine):
    """The asynchronous entry reports the flags; kbest_resolve_ties_dev with KBEST_FLAG_REFERENCE_TIES replaces the flagged problems'
    device tables by the reference-order kernel's.  The multi-device batch entry does the same by itself, before its exchange: both
    devices' global tables agree and are the checker's."""
    import torch
    E = pk.engine
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    B, N, M, k = 48, 16, 16, 90
    costs = rng.random((B, N * M)) * 7
    costs[: B // 2] = rng.integers(0, 7, size=(B // 2, N * M)).astype(np.float64)
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(costs, N, M, k)
    d_cost = torch.from_numpy(costs).to(dev)
    d_r = torch.zeros((B, k, M), dtype=torch.int32, device=dev)
    d_c = torch.zeros((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.zeros((B, k), dtype=torch.float64, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    engine.reserve(B, N, k)
    engine.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=st, d_tie_flags=d_f)
    engine.resolve_ties_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_f, stream=st, reference_ties=True)
    torch.cuda.synchronize()
    fl = d_f.cpu().numpy()
    assert (d_n.cpu().numpy() == onf).all() and (bits(d_g.cpu().numpy()) == bits(og)).all()
    assert (d_r.cpu().numpy() == or4c).all() and (d_c.cpu().numpy() == oc4r).all()
    assert (fl[B // 2:] == 0).all() and (fl[: B // 2] & E.KBEST_TIE_REFERENCE).any()
    multi = pk.KBestMulti([0, 0])
    nf, r4c, c4r, g = multi.kbest(costs, N, M, k, reference_ties=True)
    assert multi.tables_agree()
    mfl = multi.last_tie_flags()
    multi.close()
    assert (nf == onf).all() and (bits(g) == bits(og)).all() and (r4c == or4c).all() and (c4r == oc4r).all()
    assert (mfl[B // 2:] == 0).all() and (mfl[: B // 2] & E.KBEST_TIE_REFERENCE).any()


def test_association_probabilities_reference_ties():
    """kbest_set_reference_order(ctx, 2): the fused association kernels, and only the frames whose k-th and (k+1)-th gains are equal
    again through the reference-order kernel.  The probabilities of EVERY frame are the checker's (the compiled reference's heap
    order where a level straddles slot k), rel 1e-12; some frames were run again (KBEST_TIE_REFERENCE), most were not."""
    E = pk.engine
    eng = pk.KBestEngine(0)
    eng.set_reference_order(2)
    rng = np.random.default_rng(45)
    checked = rerun = 0
    for (nL, nM, k, hi) in ((6, 3, 20, 6), (12, 5, 100, 8), (20, 10, 200, 12), (20, 10, 50, 4)):
        nR = nL + nM
        frames = []
        for i in range(40):
            C_ = np.full(nR * nM, np.inf)
            for c in range(nM):
                near = rng.random(nL) < 4.0 / nL
                near[c % nL] = True
                vals = rng.integers(0, hi, nL).astype(np.float64) if i % 2 == 0 else hi * rng.random(nL)
                C_[c * nR: c * nR + nL] = np.where(near, vals, 60 + rng.integers(0, 400, nL))
                C_[c * nR + nL + c] = 10.0
            frames.append(C_)
        P, nf = eng.weights(frames, [nL] * len(frames), [nM] * len(frames), k, condition=True)
        fl = eng.last_tie_flags()
        for f, fr in enumerate(frames):
            cond, idx = ol.condition_costs(fr, nR, nM)
            cl = len(idx) - nM
            q, n = ol.assignment_prob(cond, cl, nM, k)
            full = np.zeros((nM, nL + 1))
            full[:, np.asarray(idx[:cl], dtype=np.int64)] = q[:, :cl]
            full[:, nL] = q[:, cl]
            assert nf[f] == n, (nL, nM, k, f)
            np.testing.assert_allclose(P[f], full, rtol=1e-12, atol=1e-300, err_msg=str((nL, nM, k, f, hex(int(fl[f])))))
            assert not (fl[f] & (E.KBEST_TIE_UNRESOLVED | E.KBEST_TIE_RESOLVED))
            checked += 1
            rerun += int(bool(fl[f] & E.KBEST_TIE_REFERENCE))
    assert checked == 160 and 10 < rerun < 120
    eng.close()


def test_association_with_more_than_1024_kept_rows(engine):
    """getAssignmentProbs / assignmentProb on frames that keep more than 1 024 rows after conditionCosts (1 100 and 1 500 landmarks
    within the gate): the general pipeline with the reference-order kernel as its enumeration; the checker's probabilities."""
    rng = np.random.default_rng(1)
    for (nL, nM, k) in ((1100, 3, 20), (1500, 2, 10)):
        nR = nL + nM
        C_ = np.full(nR * nM, np.inf)
        for c in range(nM):
            C_[c * nR: c * nR + nL] = 30.0 * rng.random(nL)
            C_[c * nR + nL + c] = 10.0
        P, nf = engine.weights([C_], [nL], [nM], k, condition=True)
        assert engine.last_route() == pk.engine.KBEST_ROUTE_EXACT
        cond, idx = ol.condition_costs(C_, nR, nM)
        cl = len(idx) - nM
        assert len(idx) > 1024
        q, n = ol.assignment_prob(cond, cl, nM, k)
        full = np.zeros((nM, nL + 1))
        full[:, np.asarray(idx[:cl], dtype=np.int64)] = q[:, :cl]
        full[:, nL] = q[:, cl]
        assert nf[0] == n
        np.testing.assert_allclose(P[0], full, rtol=1e-12, atol=1e-300)
        P2, nf2 = engine.weights([cond], [cl], [nM], k)
        assert nf2[0] == n
        np.testing.assert_allclose(P2[0], q, rtol=1e-12, atol=1e-300)
