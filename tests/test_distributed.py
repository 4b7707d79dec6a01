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
    G, R, Nf = kd.gather_batch(torch.from_numpy(g), torch.from_numpy(r4c), torch.from_numpy(nf), B)
    ret[rank] = (G.numpy(), R.numpy(), Nf.numpy())
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
            G, R, Nf = ret[rank]
            assert (Nf == nf).all() and (R == r4c).all() and (G.view(np.int64) == g.view(np.int64)).all()


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
    G, R, Nf = kd.merge_subtree_topk(torch.from_numpy(np.stack(gs)), torch.from_numpy(np.stack(rs)),
                                     torch.tensor(ns, dtype=torch.int32), k)
    ret[rank] = (G.numpy(), R.numpy(), Nf.numpy(), np.stack([w[0] for w in want]), np.stack([w[1] for w in want]))
    dist.destroy_process_group()


def test_subtree_mode_world2_merges_to_global_kbest():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_subtree_worker, args=(world, port, ret), nprocs=world, join=True)
        for rank in range(world):
            G, R, Nf, wg, wr = ret[rank]
            assert (Nf == G.shape[1]).all()
            assert (G.view(np.int64) == wg.view(np.int64)).all() and (R == wr).all()


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
