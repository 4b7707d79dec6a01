"""One rank of the world-2 gloo test with the REAL engine (tests/test_gpu_round5.py): started as a child process with
RANK / WORLD_SIZE / MASTER_* in the environment.  Every rank drives GPU 0 through its own KBestEngine, solves its shard
(batch mode) or its root subtrees (subtree mode), and exchanges through probabilisticsemslam_amd/distributed.py exactly
as the one-process-per-GPU deployment does (there with backend "nccl" = RCCL); the tables every rank ends up with are
written to <out>.rank<r>.npz for the parent to compare.  SURVEY 8(e); split `shortestPathCPP.cpp:455-532`."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import probabilisticsemslam_amd as pk  # noqa: E402
from probabilisticsemslam_amd import distributed as kd  # noqa: E402
from probabilisticsemslam_amd import workloads as wl  # noqa: E402


def main():
    out, B, Bsub = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = pk.KBestEngine(0)  # the HIP engine: fails loudly without a GPU
    res = {}
    # KBEST_DIST_NARROW=1 (round 6): the engine returns int8 tables (KBEST_FLAG_TABLES_I8) and the exchange moves them as they are --
    # batch mode: int8 slices; subtree mode: gains first (all-gather of the top-k costs, sum all-reduce of the winners' rows)
    narrow = os.environ.get("KBEST_DIST_NARROW") == "1"
    # batch mode: a contiguous block of matrices per rank, ONE packed all-gather
    costs, N, M, k = wl.dense_config("c2", B=B)
    lo, hi = kd.shard_range(B, rank, world)
    nf, r4c, c4r, g = eng.kbest(costs[lo:hi], N, M, k, tables_i8=narrow)
    G, R, Nf = kd.gather_batch(torch.from_numpy(g), torch.from_numpy(r4c), torch.from_numpy(nf), B)
    assert R.dtype == (torch.int8 if narrow else torch.int32)
    res.update(batch_g=G.numpy(), batch_r=R.numpy().astype(np.int32), batch_nf=Nf.numpy())
    # subtree mode: every rank enumerates the root children on its columns (reference column order), the exchange, the k-way merge
    for tag, (cc, n, m, kk) in {"sub": wl.dense_config("c2", B=Bsub), "sub64": (wl.dense_config("c4", B=2)[0], 64, 64, 60)}.items():
        nf, r4c, c4r, g = eng.kbest(cc, n, m, kk, root_shard=(rank, world), tables_i8=narrow)
        G, R, Nf = kd.merge_subtree_topk(torch.from_numpy(g), torch.from_numpy(r4c), torch.from_numpy(nf), kk)
        res.update({f"{tag}_g": G.numpy(), f"{tag}_r": R.numpy().astype(np.int32), f"{tag}_nf": Nf.numpy(),
                    f"{tag}_path": np.array(kd.last_exchange["path"])})
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
