"""CPU, world_size 2, gloo: the sharding + exchange layer (probabilisticsemslam_amd/distributed.py) gives the
same tables as a single rank.  The per-rank solver here is the CPU checker (the GPU engine is exercised by the
-m gpu tests; test_subtree_sharding_merges_to_global_kbest covers the engine's root_shard option itself)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from probabilisticsemslam_amd import distributed as kd
from probabilisticsemslam_amd import workloads as wl


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _batch_worker(rank, world, port, B, ret):
    _init(rank, world, port)
    costs, N, M, k = wl.dense_config("c2", B=B)
    lo, hi = kd.shard_range(B, rank, world)
    nf, r4c, c4r, g, _ = ol.orc_kbest_batch(costs[lo:hi], N, M, k)
    out = []
    for max_row in (None, N):  # int32 slices, and row4col as bytes (every index of a 16-row problem fits one)
        G, R, Nf = kd.gather_batch(torch.from_numpy(g), torch.from_numpy(r4c), torch.from_numpy(nf), B, max_row=max_row)
        out.append((G.numpy(), R.numpy(), Nf.numpy()))
    ret[rank] = out
    dist.destroy_process_group()


def test_shard_range_partitions():
    for B in (0, 1, 7, 128, 1025):
        for W in (1, 2, 3, 8):
            r = [kd.shard_range(B, g, W) for g in range(W)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(W - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_batch_mode_world2_equals_single_rank():
    B, world = 13, 2   # odd: uneven shards
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_batch_worker, args=(world, port, B, ret), nprocs=world, join=True)
        costs, N, M, k = wl.dense_config("c2", B=B)
        nf, r4c, c4r, g, _ = ol.orc_kbest_batch(costs, N, M, k)
        for rank in range(world):
            for G, R, Nf in ret[rank]:
                assert (Nf == nf).all() and (R == r4c).all() and (G.view(np.int64) == g.view(np.int64)).all()
        assert kd.slice_bytes(7, k, M, True) * 2.9 < kd.slice_bytes(7, k, M, False)  # 8 + M against 8 + 4 M bytes per solution (M = 16: 24 against 72)


def _subtree_lists(cost, N, M, k, world, big):
    """Per-rank k-best of the rank's root subtrees, derived from one long checker enumeration: a solution
    belongs to the root child on the first column where it differs from the root (Murty's partition)."""
    nf, r4c, c4r, g = ol.orc_kbest(cost, N, M, big)
    root = r4c[0]
    first_diff = np.array([np.argmax(r4c[s] != root) for s in range(1, nf)])
    lists = []
    for rank in range(world):
        own = 1 + np.nonzero(first_diff % world == rank)[0][: k - 1]
        idx = np.concatenate([[0], own])
        gg = np.zeros(k); rr = np.zeros((k, M), np.int32)
        gg[: len(idx)] = g[idx]; rr[: len(idx)] = r4c[idx]
        lists.append((gg, rr, len(idx)))
    return lists, (g[:k], r4c[:k])


def _subtree_worker(rank, world, port, ret):
    _init(rank, world, port)
    costs, N, M, k = wl.dense_config("c2", B=3)
    gs, rs, ns, want = [], [], [], []
    for b in range(3):
        lists, glob = _subtree_lists(costs[b], N, M, k, world, 40 * k)
        gs.append(lists[rank][0]); rs.append(lists[rank][1]); ns.append(lists[rank][2]); want.append(glob)
    out = []
    # the whole lists as int32 (round 5's exchange), then gains first: the all-gather of the top-k costs + the sum all-reduce of the
    # winners' rows
    for max_row, path in ((None, "whole_lists"), (N, "gains_first")):
        G, R, Nf = kd.merge_subtree_topk(torch.from_numpy(np.stack(gs)), torch.from_numpy(np.stack(rs)),
                                         torch.tensor(ns, dtype=torch.int32), k, max_row=max_row)
        assert kd.last_exchange["path"] == path, kd.last_exchange
        out.append((G.numpy(), R.numpy(), Nf.numpy(), kd.last_exchange["bytes_sent"]))
    ret[rank] = (out, np.stack([w[0] for w in want]), np.stack([w[1] for w in want]))
    dist.destroy_process_group()


def test_subtree_mode_world2_merges_to_global_kbest():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_subtree_worker, args=(world, port, ret), nprocs=world, join=True)
        for rank in range(world):
            out, wg, wr = ret[rank]
            for G, R, Nf, sent in out:
                assert (Nf == G.shape[1]).all()
                assert (G.view(np.int64) == wg.view(np.int64)).all() and (R == wr).all()
            assert out[1][3] * 2 < out[0][3]  # gains first: 8 k + k M bytes per matrix against (8 + 4 M) k


def _tie_worker(rank, world, port, ret):
    """Integer costs: exactly equal gains across the ranks' lists.  The gains-first exchange must notice (the same gathered gains on
    every rank) and fall back to the whole lists, whose merge orders ties by the assignment: the table is the single-list merge's."""
    _init(rank, world, port)
    rng = np.random.default_rng(5)
    N = M = 8
    k = 40
    costs = np.floor(rng.random((4, N * M)) * 4)
    gs, rs, ns = [], [], []
    for b in range(4):
        lists, _ = _subtree_lists(costs[b], N, M, k, world, 3000)
        gs.append(lists[rank][0]); rs.append(lists[rank][1]); ns.append(lists[rank][2])
    tg, tr, tn = torch.from_numpy(np.stack(gs)), torch.from_numpy(np.stack(rs)), torch.tensor(ns, dtype=torch.int32)
    a = kd.merge_subtree_topk(tg, tr, tn, k, max_row=N)
    path_a = kd.last_exchange["path"]
    b_ = kd.merge_subtree_topk(tg, tr, tn, k)  # int32 whole lists
    ret[rank] = (path_a, [x.numpy() for x in a], [x.numpy() for x in b_])
    dist.destroy_process_group()


def test_subtree_mode_exact_ties_fall_back_to_the_whole_lists():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tie_worker, args=(world, port, ret), nprocs=world, join=True)
        for rank in range(world):
            path, a, b = ret[rank]
            assert path == "whole_lists"
            assert all((x == y).all() for x, y in zip(a, b))
        assert all((x == y).all() for x, y in zip(ret[0][1], ret[1][1]))  # identical on both ranks


def test_merge_gains_positions_match_the_list_merge():
    """merge_gains (no assignments) against merge_lists on tie-free random lists: same gains, same counts, and scattering every
    shard's rows to the positions it reports rebuilds merge_lists' table -- what the sum all-reduce does across ranks."""
    rng = np.random.default_rng(3)
    for (W, B, kk, M, k, maximize) in ((3, 5, 20, 6, 20, False), (8, 2, 50, 4, 50, True), (2, 3, 10, 3, 7, False), (4, 2, 6, 5, 30, False)):
        G = np.sort(rng.random((W, B, kk)), axis=2)
        if maximize:
            G = G[:, :, ::-1].copy()
        G[:, :, 0] = G[0:1, :, 0]  # slot 0: the root, the same on every shard
        R = rng.integers(0, 100, (W, B, kk, M)).astype(np.int32)
        R[:, :, 0] = R[0:1, :, 0]
        Nf = rng.integers(1, kk + 1, (W, B)).astype(np.int32)
        Nf[:, 0] = 0  # an infeasible matrix: every shard solves the same root
        tG, tR, tN = torch.from_numpy(G), torch.from_numpy(R), torch.from_numpy(Nf)
        wg, wr, wn = kd.merge_lists(tG, tR, tN, k, maximize)
        pos, og, on, tied = kd.merge_gains(tG, tN, k, maximize)
        assert not tied
        assert (on.numpy()[1:] == wn.numpy()[1:]).all() and on[0] == 0
        table = np.zeros((B, k, M), np.int32)
        for w in range(W):
            bi, si = np.nonzero(pos[w].numpy() >= 0)
            table[bi, pos[w].numpy()[bi, si]] += R[w, bi, si]
        table[:, 0] = R[0, :, 0]
        for b in range(1, B):
            n = int(on[b])
            assert (og[b, :n].numpy().view(np.int64) == wg[b, :n].numpy().view(np.int64)).all()
            assert (table[b, :n] == wr[b, :n].numpy()).all()


def test_bench_launcher_command_and_fail_fast(capsys):
    """`python bench.py --gpus N` (no WORLD_SIZE) turns itself into the torch.distributed.run line the bench contract names,
    as a child process; with too few GPUs it returns 2 without spawning anything."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"], port=29600)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29600"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]
    assert bench.self_launch(4, ["--gpus", "4"], count=lambda: 1) == 2
    assert "nothing launched" in capsys.readouterr().err
