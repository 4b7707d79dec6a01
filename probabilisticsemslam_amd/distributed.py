"""Multi-GPU layer of the k-best path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU node, "gloo" in the CPU tests).  SURVEY 8(e).

The path shards two ways, both without any collective inside the solve:

* batch mode (default): the B cost matrices are independent, rank g solves the contiguous block
  ``shard_range(B, g, G)``.  The only exchange is the final all-gather of the per-rank result tables
  (``gather_batch``) so that every rank holds the global table.
* subtree mode (few large matrices): Murty's partition of the root is disjoint, so rank g expands only the
  root children on columns c with c % G == g (``root_shard=(g, G)`` of the engine) and enumerates its own
  k best; the global k best are the k smallest of {root} U all per-rank lists (``merge_subtree_topk``):
  one all-gather of (gain[k], row4col[k, M], nf) per matrix, then a k-way merge.

Nothing here computes assignments: tensors come from the engine (GPU) or, in the CPU tests, from the checker.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int):
    """Contiguous block of problems owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather(t: torch.Tensor, world: int) -> torch.Tensor:
    t = t.contiguous()
    if t.dim() == 0:
        t = t.view(1)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)  # concatenated along dim 0 (the layout gloo and nccl both accept)
    return out.view((world,) + tuple(t.shape))


def _pack(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor) -> torch.Tensor:
    """One rank's result tables as ONE buffer of bytes: gain fp64 | row4col i32 | nf i32 (the exchange is a single
    all-gather of the packed per-rank slices, as in bench.py and kbest_multi.cpp)."""
    parts = [gain.contiguous().to(torch.float64).view(torch.uint8).reshape(-1),
             row4col.contiguous().to(torch.int32).view(torch.uint8).reshape(-1),
             nf.contiguous().to(torch.int32).view(torch.uint8).reshape(-1)]
    return torch.cat(parts)


def _unpack(buf: torch.Tensor, world: int, n: int, k: int, M: int):
    """The gathered slices back as (gain[W, n, k], row4col[W, n, k, M], nf[W, n])."""
    per = buf.numel() // world
    b = buf.view(world, per)
    o1, o2 = n * k * 8, n * k * 8 + n * k * M * 4
    G = b[:, :o1].contiguous().view(torch.float64).view(world, n, k)
    R = b[:, o1:o2].contiguous().view(torch.int32).view(world, n, k, M)
    N = b[:, o2:o2 + n * 4].contiguous().view(torch.int32).view(world, n)
    return G, R, N


def gather_batch(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor, B: int):
    """Batch mode: every rank passes its own shard; ONE all-gather of the packed slices (padded to the largest shard); returns
    the global (gain[B,k], row4col[B,k,M], nf[B]) on every rank."""
    world = dist.get_world_size()
    per = max(shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world))
    k, M = gain.shape[1], row4col.shape[2]

    def pad(t):
        if t.shape[0] == per:
            return t
        z = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, z], 0)

    G, R, N = _unpack(_all_gather(_pack(pad(gain), pad(row4col), pad(nf)), world).reshape(-1), world, per, k, M)
    parts = [(shard_range(B, r, world)[1] - shard_range(B, r, world)[0]) for r in range(world)]
    cat = lambda X: torch.cat([X[r, :parts[r]] for r in range(world)], 0)  # noqa: E731
    return cat(G).to(gain.dtype), cat(R).to(row4col.dtype), cat(N).to(nf.dtype)


def merge_subtree_topk(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor, k: int, maximize: bool = False):
    """Subtree mode.  gain[B,k], row4col[B,k,M], nf[B] are this rank's k best of ITS root subtrees; slot 0 is
    the root itself on every rank.  Returns the global (gain[B,k], row4col[B,k,M], nf[B]), identical on all
    ranks: root first, then the k-1 best of the union of the ranks' slots 1.. in increasing cost
    (decreasing profit when maximize).  (On the GPU node the merge itself runs on the device:
    kbest_merge_topk_f64_dev / kbest_batch_f64_multi_ex; this torch form is the same rule, used with gloo.)"""
    world = dist.get_world_size()
    B, kk, M = row4col.shape
    # ONE all-gather of the packed (gain | row4col | nf) lists, then the merge: G [W, B, k], R [W, B, k, M], Nf [W, B]
    G, R, Nf = _unpack(_all_gather(_pack(gain, row4col, nf), world).reshape(-1), world, B, kk, M)
    return merge_lists(G.to(gain.dtype), R.to(row4col.dtype), Nf.to(nf.dtype), k, maximize)


def merge_lists(G: torch.Tensor, R: torch.Tensor, Nf: torch.Tensor, k: int, maximize: bool = False):
    """The k-way merge of W shards' lists G[W,B,k], R[W,B,k,M], Nf[W,B] (no communication)."""
    W, B, kk = G.shape
    bad = float("-inf") if maximize else float("inf")
    slot = torch.arange(kk, device=G.device).view(1, 1, kk)
    valid = (slot >= 1) & (slot < Nf.unsqueeze(-1))
    cand = torch.where(valid, G, torch.full_like(G, bad)).permute(1, 0, 2).reshape(B, W * kk)
    rows = R.permute(1, 0, 2, 3).reshape(B, W * kk, -1)
    # Exact ties in gain (integer-like costs; the reference's own order is a heap artefact) are
    # ordered by the assignment itself -- lexicographic row4col, the rule of the device merge (kbest_merge.hip) -- so the
    # merged table does not depend on which rank a hypothesis came from or on the number of ranks: stable sorts by the
    # columns from the last to the first (a lexicographic sort), then the stable sort by gain.
    o1 = torch.arange(W * kk, device=G.device).unsqueeze(0).expand(B, -1).contiguous()
    for c in range(rows.shape[-1] - 1, -1, -1):
        key = torch.gather(rows[..., c].to(torch.int64), 1, o1)
        o1 = torch.gather(o1, 1, torch.argsort(key, dim=1, stable=True))
    o2 = torch.argsort(torch.gather(cand, 1, o1), dim=1, descending=maximize, stable=True)
    order = torch.gather(o1, 1, o2)[:, : k - 1]
    cg = torch.gather(cand, 1, order)
    cr = torch.gather(rows, 1, order.unsqueeze(-1).expand(-1, -1, rows.shape[-1]))
    out_g = torch.cat([G[0, :, :1], cg], 1)
    out_r = torch.cat([R[0, :, :1], cr], 1)
    n_other = valid.sum(dim=(0, 2))
    feasible = Nf[0] > 0
    out_nf = torch.where(feasible, torch.clamp(1 + n_other, max=k), torch.zeros_like(n_other))
    return out_g, out_r, out_nf.to(Nf.dtype)
