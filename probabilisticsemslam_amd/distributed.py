"""Multi-GPU layer of the k-best path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU node, "gloo" in the CPU tests).  SURVEY 8(e).

The path shards two ways, both without any collective inside the solve:

* batch mode (default): the B cost matrices are independent, rank g solves the contiguous block
  ``shard_range(B, g, G)``.  The only exchange is the final all-gather of the per-rank result tables
  (``gather_batch``) so that every rank holds the global table.
* subtree mode (few large matrices): Murty's partition of the root is disjoint, so rank g expands only the
  root children on columns c with c % G == g (``root_shard=(g, G)`` of the engine) and enumerates its own
  k best; the global k best are the k smallest of {root} U all per-rank lists (``merge_subtree_topk``):
  one all-gather of the per-rank top-k COSTS (gain[k], nf) per matrix into the global k-best heap, then one sum
  all-reduce of the winners' rows (whole lists only where exactly equal gains need the assignments to be ordered).

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


def _narrow(row4col: torch.Tensor, max_row) -> bool:
    """Does row4col travel as bytes?  Every index of a problem of up to 127 rows fits an int8 (8 + M instead of 8 + 4 M bytes
    per solution).  All ranks must decide alike: by the tables' dtype (int8: what the engine returns with tables_i8) or by
    the caller's max_row, never by the values."""
    return row4col.dtype == torch.int8 or (max_row is not None and int(max_row) <= 127)


def _pack(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor, narrow: bool = False) -> torch.Tensor:
    """One rank's result tables as ONE buffer of bytes: gain fp64 | row4col (int8 when `narrow`, else i32) | nf i32 (the
    exchange is a single all-gather of the packed per-rank slices, as in bench.py and kbest_multi.cpp)."""
    parts = [gain.contiguous().to(torch.float64).view(torch.uint8).reshape(-1),
             row4col.contiguous().to(torch.int8 if narrow else torch.int32).view(torch.uint8).reshape(-1),
             nf.contiguous().to(torch.int32).view(torch.uint8).reshape(-1)]
    return torch.cat(parts)


def _unpack(buf: torch.Tensor, world: int, n: int, k: int, M: int, narrow: bool = False):
    """The gathered slices back as (gain[W, n, k], row4col[W, n, k, M], nf[W, n])."""
    per = buf.numel() // world
    b = buf.view(world, per)
    esz = 1 if narrow else 4
    o1, o2 = n * k * 8, n * k * 8 + n * k * M * esz
    G = b[:, :o1].contiguous().view(torch.float64).view(world, n, k)
    R = b[:, o1:o2].contiguous().view(torch.int8 if narrow else torch.int32).view(world, n, k, M)
    N = b[:, o2:o2 + n * 4].contiguous().view(torch.int32).view(world, n)
    return G, R, N


def slice_bytes(n: int, k: int, M: int, narrow: bool) -> int:
    """Bytes of one rank's packed slice (what it contributes to the all-gather)."""
    return n * k * 8 + n * k * M * (1 if narrow else 4) + n * 4


def gather_batch(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor, B: int, max_row=None):
    """Batch mode: every rank passes its own shard; ONE all-gather of the packed slices (padded to the largest shard); returns
    the global (gain[B,k], row4col[B,k,M], nf[B]) on every rank.  max_row (the problems' row count, the same on every rank): up
    to 127 the row4col part of the slices travels as bytes."""
    world = dist.get_world_size()
    per = max(shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world))
    k, M = gain.shape[1], row4col.shape[2]
    narrow = _narrow(row4col, max_row)

    def pad(t):
        if t.shape[0] == per:
            return t
        z = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, z], 0)

    G, R, N = _unpack(_all_gather(_pack(pad(gain), pad(row4col), pad(nf), narrow), world).reshape(-1), world, per, k, M, narrow)
    parts = [(shard_range(B, r, world)[1] - shard_range(B, r, world)[0]) for r in range(world)]
    cat = lambda X: torch.cat([X[r, :parts[r]] for r in range(world)], 0)  # noqa: E731
    return cat(G).to(gain.dtype), cat(R).to(row4col.dtype), cat(N).to(nf.dtype)


last_exchange = {"path": None, "bytes_sent": 0}  # of this process' last merge_subtree_topk (tests, bench)


def merge_subtree_topk(gain: torch.Tensor, row4col: torch.Tensor, nf: torch.Tensor, k: int, maximize: bool = False, max_row=None):
    """Subtree mode.  gain[B,k], row4col[B,k,M], nf[B] are this rank's k best of ITS root subtrees; slot 0 is
    the root itself on every rank.  Returns the global (gain[B,k], row4col[B,k,M], nf[B]), identical on all
    ranks: root first, then the k-1 best of the union of the ranks' slots 1.. in increasing cost
    (decreasing profit when maximize).

    The exchange is GAINS FIRST when row4col fits bytes (max_row <= 127, or int8 tables) -- the north star's "allgather of per-rank
    top-k costs into a global k-best heap": ONE all-gather of (gain[k], nf) per matrix, the merge of the gains on every rank
    (which tells a rank which of ITS OWN assignments made the cut), then ONE sum all-reduce of a byte table that holds every
    winner's row at its merged position (k M bytes per matrix, whatever the number of ranks).  Two candidates with exactly the
    same gain can only be ordered by their assignments: such a call -- every rank sees it in the same gathered gains -- and
    problems of more than 127 rows all-gather the whole lists and merge those (merge_lists), as in round 5.
    (On the GPU node the same two steps run on the device: kbest_merge_gains_f64_dev / kbest_batch_f64_multi_ex.)"""
    world = dist.get_world_size()
    rank = dist.get_rank()
    B, kk, M = row4col.shape
    narrow = _narrow(row4col, max_row)
    if narrow:
        head = torch.cat([gain.contiguous().to(torch.float64).view(torch.uint8).reshape(-1),
                          nf.contiguous().to(torch.int32).view(torch.uint8).reshape(-1)])
        buf = _all_gather(head, world)
        G = buf[:, : B * kk * 8].contiguous().view(torch.float64).view(world, B, kk)
        Nf = buf[:, B * kk * 8:].contiguous().view(torch.int32).view(world, B)
        pos, out_g, out_nf, tied = merge_gains(G, Nf, k, maximize)
        sent = head.numel()
        if not tied:
            table = torch.zeros((B, k, M), dtype=torch.int8, device=row4col.device)
            mine = pos[rank]                                   # [B, kk]: merged slot of each of my candidates, -1 = not among the k best
            bi, si = torch.nonzero(mine >= 0, as_tuple=True)
            table[bi, mine[bi, si]] = row4col[bi, si].to(torch.int8)
            if rank == 0:                                      # the root's row: slot 0 of shard 0
                table[:, 0] = row4col[:, 0].to(torch.int8)
            table[Nf[0] <= 0] = 0                              # (an infeasible matrix: nothing to report)
            dist.all_reduce(table, op=dist.ReduceOp.SUM)       # every entry is non-zero on at most one rank: the sum IS the table
            last_exchange.update(path="gains_first", bytes_sent=sent + table.numel())
            out_r = table.to(row4col.dtype)
            slot = torch.arange(k, device=out_r.device).view(1, k, 1)
            out_r = torch.where(slot < out_nf.view(B, 1, 1), out_r, torch.full_like(out_r, -1))  # (the engine's convention for unused slots)
            return out_g.to(gain.dtype), out_r, out_nf.to(nf.dtype)
    # the whole lists: ONE all-gather of the packed (gain | row4col | nf) slices, then the merge: G [W, B, k], R [W, B, k, M], Nf [W, B]
    packed = _pack(gain, row4col, nf, narrow)
    G, R, Nf = _unpack(_all_gather(packed, world).reshape(-1), world, B, kk, M, narrow)
    last_exchange.update(path="whole_lists", bytes_sent=packed.numel() + (B * kk * 8 + B * 4 if narrow else 0))
    return merge_lists(G.to(gain.dtype), R.to(row4col.dtype), Nf.to(nf.dtype), k, maximize)


def merge_gains(G: torch.Tensor, Nf: torch.Tensor, k: int, maximize: bool = False):
    """The merge of W shards' GAINS G[W,B,kk], Nf[W,B] (no communication, no assignments): pos[W,B,kk] = merged slot (1 .. k-1) of
    every candidate that is among the k best, -1 otherwise; the merged gains [B,k] and counts [B]; and whether two candidates of
    some matrix have exactly the same gain within the k best or at slot k (their order would be the assignments')."""
    W, B, kk = G.shape
    bad = float("-inf") if maximize else float("inf")
    slot = torch.arange(kk, device=G.device).view(1, 1, kk)
    valid = (slot >= 1) & (slot < Nf.unsqueeze(-1))
    cand = torch.where(valid, G, torch.full_like(G, bad)).permute(1, 0, 2).reshape(B, W * kk)
    order = torch.argsort(cand, dim=1, descending=maximize, stable=True)   # ties: by (shard, slot) -- only reported, never used
    sg = torch.gather(cand, 1, order)
    n_other = valid.sum(dim=(0, 2))
    feasible = Nf[0] > 0
    out_nf = torch.where(feasible, torch.clamp(1 + n_other, max=k), Nf[0].to(n_other.dtype))
    # exact ties among the winners, or between the last winner and the first loser
    top = min(k, W * kk)   # positions 0 .. k-1 of the sorted candidates = slots 1 .. k (slot k: the first loser)
    eq = (sg[:, 1:top] == sg[:, : top - 1]) & (sg[:, 1:top] != bad)
    tied = bool(eq.any().item()) if top > 1 else False
    rank_of = torch.empty_like(order)
    rank_of.scatter_(1, order, torch.arange(W * kk, device=G.device).unsqueeze(0).expand(B, -1).contiguous())
    pos = (1 + rank_of).view(B, W, kk).permute(1, 0, 2)
    ok = valid & (pos < k) & feasible.view(1, B, 1)
    pos = torch.where(ok, pos, torch.full_like(pos, -1))
    out_g = torch.zeros((B, k), dtype=G.dtype, device=G.device)
    out_g[:, 0] = G[0, :, 0]
    take = min(k - 1, W * kk)
    out_g[:, 1:1 + take] = sg[:, :take]
    out_g = torch.where(torch.arange(k, device=G.device).view(1, k) < out_nf.view(B, 1), out_g, torch.zeros_like(out_g))
    return pos, out_g, out_nf.to(Nf.dtype), tied


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
