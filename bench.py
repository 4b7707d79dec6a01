#!/usr/bin/env python3
"""bench.py -- k-best assignments/sec of the MI355X engine on BASELINE.json's workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (one batched kernel launch) over one batch of synthetic cost
matrices that already live in HBM.  Workload (config.workload): BASELINE.json configs[3] shape --
1024 dense 64x64 cost matrices, k = 200 -- PER GPU (weak scaling: every rank gets its own 1024
matrices of the same seeded stream; no data-path collective is needed because the matrices are
independent; with N > 1 each step ends with the RCCL all-gather of the per-rank top-k gains that
assembles the global result table).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch (SURVEY 8(d)
B_alg, with P counted by the engine itself in its no-prune mode and cross-checked against the oracle
on the cpu_baseline sample) / average kernel duration measured with HIP events on the launch stream.
`cpu_baseline` = the reference solver (oracle/_ref, built from the unmodified reference source with
its own -Ofast) timed on one host core on a bounded sample of the same workload; falls back to the
oracle restatement ("port") when oracle/_ref is absent.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(N, M, k, nf, pushed):
    """B_alg = 8NM + nf(8+4N+4M) + (P + nf - 1) * state(D)."""
    D = N
    state = 16 * D + 2 * D + (D + 7) // 8 + 16
    nf = np.asarray(nf, dtype=np.int64)
    pushed = np.asarray(pushed, dtype=np.int64)
    return 8 * N * M * len(nf) + int((nf * (8 + 4 * N + 4 * M)).sum()) + int(((pushed + nf - 1) * state).sum())


def cpu_baseline(costs, N, M, k, sample):
    """Reference solver on ONE host core over `sample` problems of the same batch."""
    import oracle_lib as ol
    sample = min(sample, costs.shape[0])
    c = np.ascontiguousarray(costs[:sample])
    if os.path.exists(ol.REF_OFAST_SO):
        lib = ol.ref(ofast=True)
        c4r = np.empty(sample * k * N, np.int64)
        r4c = np.empty(sample * k * M, np.int64)
        g = np.empty(sample * k)
        nf = np.empty(sample, np.int64)
        t0 = time.perf_counter()
        total = lib.ref_kbest2d_batch(sample, k, N, M, 0, c.reshape(-1), c4r, r4c, g, nf)
        dt = time.perf_counter() - t0
        kind = "reference"
        pushed = None
    else:
        t0 = time.perf_counter()
        nf, r4c, c4r, g, pushed = ol.orc_kbest_batch(c, N, M, k)
        dt = time.perf_counter() - t0
        total = int(nf.sum())
        kind = "port"
    return {"value": total / dt, "unit": "assignments/s", "cores": 1, "kind": kind,
            "sample": f"first {sample} of the {costs.shape[0]} {N}x{M} k={k} matrices of rank 0's batch, "
                      f"one kBest2D call each, single thread, {dt:.1f} s"}, r4c, g, pushed


def cpu_all_cores(costs, N, M, k):
    """The same reference solver on every host core (one thread per core, the batch cut into contiguous chunks).  The
    reference itself is single-threaded; this is the fairest host number, reported next to the 1-core baseline."""
    import concurrent.futures as cf
    import oracle_lib as ol
    if not os.path.exists(ol.REF_OFAST_SO):
        return None
    lib = ol.ref(ofast=True)
    T = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:  # a cgroup CPU quota below the visible core count: more runnable threads than that only get throttled
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            T = max(1, min(T, int(q[0]) // int(q[1])))
    except Exception:
        pass
    B = costs.shape[0]
    T = min(T, B)
    bounds = [B * i // T for i in range(T + 1)]
    c = np.ascontiguousarray(costs)

    def work(i):
        lo, hi = bounds[i], bounds[i + 1]
        n = hi - lo
        if n == 0:
            return 0
        c4r = np.empty(n * k * N, np.int64)
        r4c = np.empty(n * k * M, np.int64)
        g = np.empty(n * k)
        nf = np.empty(n, np.int64)
        return int(lib.ref_kbest2d_batch(n, k, N, M, 0, c[lo:hi].reshape(-1), c4r, r4c, g, nf))  # ctypes drops the GIL

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=T) as ex:
        total = sum(ex.map(work, range(T)))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "assignments/s", "cores": T, "kind": "reference",
            "sample": f"all {B} matrices of rank 0's batch, one kBest2D call each, {T} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4"])
    ap.add_argument("--batch", type=int, default=None, help="matrices per GPU (default: the config's B)")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import probabilisticsemslam_amd as pk
    from probabilisticsemslam_amd import workloads as wl

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # KBEST_BENCH_FORCE_DIST=1 runs the RCCL code path (init, all-gather, all-reduce) even with one rank: the only
    # way to exercise it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("KBEST_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    Bc, N, M, k, seed = wl.DENSE_CONFIGS[args.config]
    B = args.batch or Bc
    costs = wl.dense_batch(B, N, M, seed, first=rank * B)  # rank-private slice of the one seeded stream
    d_cost = torch.from_numpy(costs).to(dev)
    d_r4c = torch.empty((B, k, M), dtype=torch.int32, device=dev)
    d_c4r = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_gain = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_nf = torch.empty(B, dtype=torch.int32, device=dev)
    d_pushed = torch.zeros(B, dtype=torch.int64, device=dev)
    d_allgain = torch.empty((world * B, k), dtype=torch.float64, device=dev) if use_dist else None

    eng = pk.KBestEngine(local)
    eng.reserve(B, N, k)
    # a dedicated (non-null) HIP stream: the kernel, the timing events and the collective all go through it
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    torch.cuda.synchronize()  # the allocations / fills above ran on the default stream
    # untimed: the reference's push count P per matrix (no-prune mode), for the algorithmic byte count
    eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, d_gain, d_nf, d_pushed=d_pushed, prune=False, stream=stream)
    torch.cuda.synchronize()
    pushed = d_pushed.cpu().numpy()
    nf_ref = d_nf.cpu().numpy().copy()
    g_ref = d_gain.cpu().numpy().copy()

    # Multi-GPU: the per-rank top-k gains of step i are all-gathered (RCCL over xGMI) WHILE the kernel of step i+1
    # runs: two gain tables alternate, and a table is only written again once the gather that read it has finished
    # (stream-level wait on its work handle).  Every gather completes inside the timed region.
    d_gain2 = torch.empty_like(d_gain) if use_dist else None
    d_allgain2 = torch.empty_like(d_allgain) if use_dist else None
    gains = (d_gain, d_gain2)
    allgains = (d_allgain, d_allgain2)
    pending = [None, None]

    def step(i, ev=None):
        b = i & 1 if use_dist else 0
        if pending[b] is not None:
            pending[b].wait()  # the gather of step i-2 has read gains[b]
            pending[b] = None
        if ev is not None:
            ev[0].record()
        eng.kbest_dev(d_cost, B, N, M, k, d_r4c, d_c4r, gains[b], d_nf, stream=stream)
        if ev is not None:
            ev[1].record()
        if use_dist:
            pending[b] = dist.all_gather_into_tensor(allgains[b], gains[b], async_op=True)

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, ev[i])
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist and ((args.steps - 1) & 1) == 1:
        d_gain, d_allgain = d_gain2, d_allgain2  # the last step wrote (and gathered) the second pair of tables
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    nf = d_nf.cpu().numpy()
    g = d_gain.cpu().numpy()
    # the timed (pruning) path must reproduce the no-prune run bit for bit
    parity_self = bool((nf == nf_ref).all() and (g.view(np.int64) == g_ref.view(np.int64)).all())
    found = int(nf.sum())
    balg = algorithmic_bytes(N, M, k, nf, pushed)

    t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
    tot = torch.tensor([found, balg], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max, kern_ms_max = float(t[0]), float(t[1])
    found_all, balg_all = float(tot[0]), float(tot[1])

    if rank == 0:
        out = {
            "metric": "k-best assignments/sec (batched NxN cost matrices, k=200)",
            "value": found_all * args.steps / dt_max,
            "unit": "assignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} dense {N}x{M} cost matrices per GPU, k={k} (BASELINE configs[3] shape, "
                                   f"splitmix64 seed {seed:#x}), kBest2D semantics", "matrices_per_gpu": B,
                       "numRow": N, "numCol": M, "k": k, "parallelism": f"batch-sharded x{world}"},
            "problems_per_s": B * world * args.steps / dt_max,
            "kernel_ms": kern_ms_max,
            "parity_prune_vs_noprune": parity_self,
            "roofline": {"bound": "hbm", "achieved": (balg_all / world) / (kern_ms_max * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (balg_all / world) / (kern_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "algorithmic_bytes_per_launch": balg_all / world,
                         "mean_pushed_per_matrix": float(pushed.mean())},
        }
        tr = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tr):  # PMC-measured HBM bytes per launch of this workload (see profiles/README.md)
            try:
                j = json.load(open(tr))
                if j.get("config") == args.config and j.get("batch") == B:
                    out["roofline"]["traffic"] = j["bytes_per_launch"]
            except Exception:
                pass
        if world == 1 and not args.no_cpu:
            sample = args.cpu_sample or {"c4": 512, "c3": 2048, "c2": 1024}[args.config]
            cb, r4c_cpu, g_cpu, p_cpu = cpu_baseline(costs, N, M, k, sample)
            ns = min(sample, B)
            cb["parity_vs_gpu"] = bool((g_cpu.reshape(-1, k)[:ns].view(np.int64) == g[:ns].view(np.int64)).all()
                                       and (np.asarray(r4c_cpu).reshape(-1, k, M)[:ns] == d_r4c.cpu().numpy()[:ns]).all())
            if p_cpu is not None:
                cb["pushed_matches_gpu"] = bool((p_cpu[:ns] == pushed[:ns]).all())
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1core"] = out["value"] / cb["value"]
            ca = cpu_all_cores(costs, N, M, k)
            if ca is not None:
                out["cpu_baseline_all_cores"] = ca
                out["speedup_vs_cpu_all_cores"] = out["value"] / ca["value"]
        line = json.dumps(out)
    if use_dist:
        if world > 1:  # every rank must hold the same global table
            assert torch.equal(d_allgain[rank * B:(rank + 1) * B], d_gain)
        dist.destroy_process_group()
    # RCCL prints a banner through C stdio, which is flushed at exit, i.e. AFTER anything Python has printed: push
    # it out first so that the JSON line is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
