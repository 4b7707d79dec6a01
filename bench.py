#!/usr/bin/env python3
"""bench.py -- k-best assignments/sec of the MI355X engine on BASELINE.json's workloads.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (WORLD_SIZE unset: launches the line above itself, as a child process, before
                                         anything touches the GPU; rc 2 and one line if the node has fewer than N GPUs)

One "step" = one pass of the hot path (one batched kernel launch) over one batch of synthetic inputs that already
live in HBM.  Headline workload (config.workload): BASELINE.json configs[3] shape -- 1024 dense 64x64 cost matrices,
k = 200.  `--scaling weak` (default): PER GPU (every rank gets its own 1024 matrices of the same seeded stream);
`--scaling strong`: configs[3] literally -- 1024 matrices in all, a contiguous block per rank.  The matrices are
independent, so there is no data-path collective; with N > 1 each step ends with the RCCL all-gather of the packed
per-rank result table (gain[k], row4col[k*M], nf per matrix -- SURVEY 8(e)) that assembles the global table, overlapped
with the next step's kernel.

Prints ONE JSON line (rank 0):
  roofline.achieved   = algorithmic bytes per launch (SURVEY 8(d) B_alg, with P counted by the engine itself in its
                        no-prune mode and cross-checked against the reference on the cpu_baseline sample) / average
                        kernel duration measured with HIP events on the launch stream; roofline.traffic = HBM bytes per
                        launch from the committed rocprofv3 PMC passes (profiles/);
  issue               = what actually bounds the kernel (vector-unit busy fraction etc.) from the same profiles;
  cpu_baseline        = the reference solver (oracle/_ref, built from the unmodified reference source with its own
                        -Ofast) timed on one host core on a bounded sample of the same workload; "port" (the oracle
                        restatement) when oracle/_ref is absent;
  value_host_inclusive= the same batch through the host-pointer entry (H2D of the cost blocks and D2H of the result
                        tables included -- SURVEY 8(d)'s metric definition; never `value`);
  configs             = the other BASELINE configs measured in the same run: c2 (1024 x 16x16, k=50), c3 (4096 x
                        32x32, k=200), c5 (1000 streamed 30x10 KITTI-like frames through the fused association kernel:
                        batched throughput AND one-frame-per-call latency, weights checked against the reference).
`--config c2|c3|c5` makes that configuration the headline of the line instead.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9       # peak shader clock (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs
N_SIMD = 1024


def state_bytes(D):
    """one hypothesis = u, v (fp64), row4col, col4row (u8), forbidden bitmask, gain/activeCol/flags."""
    return 16 * D + 2 * D + (D + 7) // 8 + 16


def algorithmic_bytes(N, M, k, nf, pushed):
    """B_alg = 8NM + nf(8+4N+4M) + (P + nf - 1) * state(D)."""
    nf = np.asarray(nf, dtype=np.int64)
    pushed = np.asarray(pushed, dtype=np.int64)
    return 8 * N * M * len(nf) + int((nf * (8 + 4 * N + 4 * M)).sum()) + int(((pushed + nf - 1) * state_bytes(N)).sum())


def profile_summary(cfg):
    """profiles/rNN_<cfg>_summary.json of the newest round that has one (tools/prof.sh + tools/prof_summary.py on the GPU box):
    HBM traffic and the instruction-issue counters of the dominant kernel for this configuration, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{cfg}_summary.json")), reverse=True):
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                j["_file"] = os.path.relpath(path, ROOT)
                return j
            except Exception:
                pass
    return None


def issue_block(cfg):
    """The bound that is real for this path: instruction issue.  VALU-busy = SQ_INSTS_VALU x 4 cycles (fp64 / 16 lanes
    per clock per SIMD) / (1024 SIMDs x kernel cycles)."""
    j = profile_summary(cfg)
    if not j or "pmc" not in j or "kernel_avg_us_timed" not in j:
        return None
    pm, us = j["pmc"], j["kernel_avg_us_timed"]
    cyc = N_SIMD * us * 1e-6 * CLOCK_HZ
    out = {"source": j["_file"], "kernel_us": us}
    if "SQ_INSTS_VALU" in pm:
        out["valu_busy_frac"] = pm["SQ_INSTS_VALU"] * 4.0 / cyc
        out["valu_insts"] = pm["SQ_INSTS_VALU"]
    if "SQ_INSTS_SALU" in pm:
        out["salu_insts"] = pm["SQ_INSTS_SALU"]
    if "SQ_INSTS_LDS" in pm:
        out["lds_insts"] = pm["SQ_INSTS_LDS"]
    if "SQ_WAIT_ANY" in pm and "SQ_WAVE_CYCLES" in pm and pm["SQ_WAVE_CYCLES"]:
        out["wave_wait_frac"] = pm["SQ_WAIT_ANY"] / pm["SQ_WAVE_CYCLES"]
    out["bound"] = "instruction issue / latency (not HBM): see DESIGN.md section 4 and NOTES.md section 8"
    return out


def roofline_block(cfg, B, balg, kern_ms, extra=None):
    ach = balg / (kern_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "achieved_basis": "algorithmic bytes per launch (SURVEY 8(d) B_alg) / mean kernel time (HIP events)",
         "traffic": None, "algorithmic_bytes_per_launch": balg}
    j = profile_summary(cfg)
    if j and j.get("batch") == B and "bytes_per_launch" in j:  # PMC-measured HBM bytes per launch (profiles/README.md)
        r["traffic"] = j["bytes_per_launch"]
        r["traffic_GBps"] = j["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
        r["traffic_frac_of_peak"] = r["traffic_GBps"] / HBM_PEAK_GBS
        r["traffic_source"] = j["_file"]
    if extra:
        r.update(extra)
    return r


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_dense(costs, N, M, k, sample):
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
    ref = (np.asarray(nf).reshape(-1), np.asarray(r4c).reshape(sample, k, M), np.asarray(c4r).reshape(sample, k, N), np.asarray(g).reshape(sample, k))
    return {"value": total / dt, "unit": "assignments/s", "cores": 1, "kind": kind,
            "sample": f"first {sample} of the {costs.shape[0]} {N}x{M} k={k} matrices of rank 0's batch, "
                      f"one kBest2D call each, single thread, {dt:.1f} s"}, ref, pushed


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

    # the reference's tables of the WHOLE batch are kept: the GPU result of every matrix is compared with them (tables_parity)
    C4R = np.empty((B, k, N), np.int64)
    R4C = np.empty((B, k, M), np.int64)
    G = np.empty((B, k))
    NF = np.empty(B, np.int64)

    def work(i):
        lo, hi = bounds[i], bounds[i + 1]
        n = hi - lo
        if n == 0:
            return 0
        return int(lib.ref_kbest2d_batch(n, k, N, M, 0, c[lo:hi].reshape(-1), C4R[lo:hi].reshape(-1), R4C[lo:hi].reshape(-1),
                                         G[lo:hi].reshape(-1), NF[lo:hi]))  # ctypes drops the GIL

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=T) as ex:
        total = sum(ex.map(work, range(T)))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "assignments/s", "cores": T, "kind": "reference",
            "sample": f"all {B} matrices of rank 0's batch, one kBest2D call each, {T} threads, {dt:.1f} s"}, (NF, R4C, C4R, G)


def tables_parity(m, ref, ns=None):
    """The timed launch's tables (nf, row4col, col4row, gain bits) against the reference's on the first ns matrices (None: all).
    col4row on zero-padded columns (values >= numCol) is compared after mapping to -1."""
    NF, R4C, C4R, G = ref
    N, M, k = m["N"], m["M"], m["k"]
    ns = len(NF) if ns is None else min(ns, len(NF))
    nf_ok = bool((m["nf"][:ns] == NF[:ns]).all())
    sl = np.arange(k)[None, :] < np.asarray(NF[:ns])[:, None]  # slots the reference filled
    g_ok = bool((np.where(sl, m["g"][:ns].view(np.int64), 0) == np.where(sl, np.asarray(G[:ns]).reshape(ns, k).view(np.int64), 0)).all())
    r_ok = bool((np.where(sl[:, :, None], m["r4c"][:ns], 0) == np.where(sl[:, :, None], np.asarray(R4C[:ns]).reshape(ns, k, M), 0)).all())
    a = m["c4r"][:ns].astype(np.int64)
    b = np.asarray(C4R[:ns]).reshape(ns, k, N).astype(np.int64)
    a = np.where(a >= M, -1, a)
    b = np.where(b >= M, -1, b)
    c_ok = bool((np.where(sl[:, :, None], a, 0) == np.where(sl[:, :, None], b, 0)).all())
    return {"nf": nf_ok, "gain_bits": g_ok, "row4col": r_ok, "col4row": c_ok, "matrices": int(ns), "all": nf_ok and g_ok and r_ok and c_ok}


# ------------------------------------------------------------------------------------------------ dense configs
class Exchange:
    """The one collective of a step: ONE all-gather of this rank's packed slice into the global table, asynchronous,
    so that it travels while the next step's kernel runs.  Backend "nccl" (= RCCL over xGMI: what the driver's multi-GPU run uses)
    gathers device buffers directly.  Backend "gloo" (KBEST_BENCH_BACKEND=gloo: how every line of the world > 1 path runs where
    RCCL cannot -- two ranks on ONE GPU, tests/test_gpu_round6.py) moves the same packed bytes through pinned host memory."""

    def __init__(self, torch, dist, backend, dev, world, slice_bytes, nbuf, stream):
        self.torch, self.dist, self.backend, self.world, self.stream = torch, dist, backend, world, stream
        self.g_pack = [torch.empty(world * slice_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        if backend == "gloo":
            self.h_pack = [torch.empty(slice_bytes, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
            self.h_gpack = [torch.empty(world * slice_bytes, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
            self.ev = [torch.cuda.Event() for _ in range(nbuf)]

    def start(self, b, d_pack):
        """Called right after step's kernel was enqueued on the stream; returns what wait() needs."""
        if self.backend != "gloo":
            return [self.dist.all_gather_into_tensor(self.g_pack[b], d_pack, async_op=True)]
        with self.torch.cuda.stream(self.stream):
            self.h_pack[b].copy_(d_pack, non_blocking=True)
            self.ev[b].record(self.stream)
        self.ev[b].synchronize()  # (the host bounce cannot start before the kernel has ended: no overlap on this backend)
        return [self.dist.all_gather_into_tensor(self.h_gpack[b], self.h_pack[b], async_op=True), b]

    def wait(self, pending):
        pending[0].wait()
        if self.backend == "gloo":
            b = pending[1]
            with self.torch.cuda.stream(self.stream):
                self.g_pack[b].copy_(self.h_gpack[b], non_blocking=True)


def all_reduce_(torch, dist, backend, t, op):
    """dist.all_reduce of a small device tensor; through the host on the gloo backend."""
    if backend == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def run_dense(eng, torch, dist, cfg, B, steps, warmup, rank, world, dev, tstream, use_dist, first, backend="nccl", no_gather_leg=False):
    """C2 / C3 / C4: B dense matrices resident in HBM, one launch per step.  Returns the measurements of this rank."""
    from probabilisticsemslam_amd import workloads as wl
    _, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
    costs = wl.dense_batch(B, N, M, seed, first=first)  # rank-private slice of the one seeded stream
    stream = tstream.cuda_stream
    d_cost = torch.from_numpy(costs).to(dev)
    # What a rank contributes to the exchange is as narrow as the problem allows: every index of a problem of up to 127 rows fits a
    # byte, so with N > 1 the kernel writes row4col (and col4row, which stays local) as int8 tables (KBEST_FLAG_TABLES_I8) --
    # 8 + M instead of 8 + 4 M bytes per solution.  One GPU, no exchange: the int32 tables of the interface.
    i8 = bool(use_dist and N <= 127 and os.environ.get("KBEST_BENCH_WIDE_SLICES") != "1")
    tdt, esz = (torch.int8, 1) if i8 else (torch.int32, 4)
    d_c4r = torch.empty((B, k, N), dtype=tdt, device=dev)
    d_pushed = torch.zeros(B, dtype=torch.int64, device=dev)
    nbuf = 2 if use_dist else 1
    # One packed slice per rank -- gain[B][k] fp64 | row4col[B][k][M] (int8 / i32) | nf[B] i32, each part 16-byte aligned -- so that
    # the exchange is ONE all-gather of bytes per step; the kernel writes straight into views of the slice.
    up16 = lambda x: (x + 15) & ~15  # noqa: E731
    off_r = up16(B * k * 8)
    off_n = off_r + up16(B * k * M * esz)
    slice_bytes = off_n + up16(B * 4)
    d_pack = [torch.zeros(slice_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_gain = [pk[: B * k * 8].view(torch.float64).view(B, k) for pk in d_pack]
    d_r4c = [pk[off_r: off_r + B * k * M * esz].view(tdt).view(B, k, M) for pk in d_pack]
    d_nf = [pk[off_n: off_n + B * 4].view(torch.int32) for pk in d_pack]
    # global result table of the all-gather: the ranks' slices one after the other
    ex = Exchange(torch, dist, backend, dev, world, slice_bytes, nbuf, tstream) if use_dist else None

    def g_view(b, r):
        s = ex.g_pack[b][r * slice_bytes: (r + 1) * slice_bytes]
        return (s[: B * k * 8].view(torch.float64).view(B, k), s[off_r: off_r + B * k * M * esz].view(tdt).view(B, k, M),
                s[off_n: off_n + B * 4].view(torch.int32))
    eng.reserve(B, N, k)
    torch.cuda.synchronize()  # the allocations / fills above ran on the default stream
    # untimed: the reference's push count P per matrix (no-prune mode), for the algorithmic byte count
    eng.kbest_dev(d_cost, B, N, M, k, d_r4c[0], d_c4r, d_gain[0], d_nf[0], d_pushed=d_pushed, prune=False, stream=stream, tables_i8=i8)
    torch.cuda.synchronize()
    pushed = d_pushed.cpu().numpy()
    nf_ref = d_nf[0].cpu().numpy().copy()
    g_ref = d_gain[0].cpu().numpy().copy()
    pending = [None] * nbuf

    def step(i, ev=None, gather=True):
        b = i % nbuf
        if pending[b] is not None:
            ex.wait(pending[b])  # the gather of step i-2 has read this set of tables
            pending[b] = None
        if ev is not None:
            ev[0].record()
        eng.kbest_dev(d_cost, B, N, M, k, d_r4c[b], d_c4r, d_gain[b], d_nf[b], stream=stream, tables_i8=i8)
        if ev is not None:
            ev[1].record()
        if use_dist and gather:  # the packed per-rank slice travels while the next step's kernel runs: one collective per step
            pending[b] = ex.start(b, d_pack[b])

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                ex.wait(pending[b])
                pending[b] = None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(gather):
        for i in range(warmup):
            step(i, gather=gather)
        drain()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, ev[i], gather=gather)
        drain()
        barrier()
        return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # the same K steps WITHOUT the exchange first (N > 1 only; untimed as far as `value` goes): the collective's exposed time is
    # the difference of the two legs
    dt_nog = kern_nog = None
    if use_dist and not no_gather_leg:
        dt_nog, kern_nog = timed(False)
    dt, kern_ms = timed(True)
    last = (steps - 1) % nbuf
    nf = d_nf[last].cpu().numpy()
    g = d_gain[last].cpu().numpy()
    r4c = d_r4c[last].cpu().numpy().astype(np.int32)
    c4r = d_c4r.cpu().numpy().astype(np.int32)
    # the timed (pruning) path must reproduce the no-prune run bit for bit
    parity_self = bool((nf == nf_ref).all() and (g.view(np.int64) == g_ref.view(np.int64)).all())
    if use_dist:  # every rank must hold the same global table, and its own slice of it must be what it solved
        mine = g_view(last, rank)
        assert torch.equal(mine[0], d_gain[last]) and torch.equal(mine[1], d_r4c[last]) and torch.equal(mine[2], d_nf[last])
    if use_dist and world > 1:
        parts = [g_view(last, r) for r in range(world)]
        chk = torch.stack([sum(p[0].sum() for p in parts), sum(p[1].double().sum() for p in parts),
                           sum(p[2].double().sum() for p in parts)])
        lo, hi = chk.clone(), chk.clone()
        all_reduce_(torch, dist, backend, lo, dist.ReduceOp.MIN)
        all_reduce_(torch, dist, backend, hi, dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks hold different global tables"
        # ... and every other rank's slice must be what THAT rank's matrices give: the neighbour's first matrix, solved here
        nb = (rank + 1) % world
        if first == rank * B:  # (how main() deals the one seeded stream of matrices: rank r starts at matrix r * B)
            c1 = torch.from_numpy(wl.dense_batch(1, N, M, seed, first=nb * B)).to(dev)
            r1 = torch.empty((1, k, M), dtype=tdt, device=dev)
            q1 = torch.empty((1, k, N), dtype=tdt, device=dev)
            g1 = torch.empty((1, k), dtype=torch.float64, device=dev)
            n1 = torch.empty(1, dtype=torch.int32, device=dev)
            eng.kbest_dev(c1, 1, N, M, k, r1, q1, g1, n1, stream=stream, tables_i8=i8)
            torch.cuda.synchronize()
            assert torch.equal(parts[nb][0][0], g1[0]) and torch.equal(parts[nb][1][0], r1[0]) and int(parts[nb][2][0]) == int(n1[0]), \
                "the neighbour's slice of the global table is not what its first matrix gives"
    coll = None
    if use_dist:
        coll = {"what": "ONE all-gather per step of the packed per-rank slice (gain[k] fp64 | row4col[k*M] "
                        f"{'int8' if i8 else 'int32'} | nf per matrix), overlapped with the next step's kernel",
                "backend": backend, "row4col_dtype": "int8" if i8 else "int32",
                "bytes_per_rank_per_step": slice_bytes, "bytes_inbound_per_rank_per_step": (world - 1) * slice_bytes,
                "inbound_GBps_per_rank_at_this_step_rate": (world - 1) * slice_bytes / (dt / steps) / 1e9,
                "ms_per_step_without_the_gather": None if dt_nog is None else 1e3 * dt_nog / steps,
                "exposed_ms": None if dt_nog is None else 1e3 * (dt - dt_nog) / steps,
                "kernel_ms_without_the_gather": kern_nog}
    return dict(costs=costs, N=N, M=M, k=k, seed=seed, B=B, dt=dt, kern_ms=kern_ms, nf=nf, g=g, r4c=r4c, c4r=c4r, pushed=pushed,
                parity_self=parity_self, found=int(nf.sum()), balg=algorithmic_bytes(N, M, k, nf, pushed), collective=coll)


def host_inclusive_dense(eng, costs, N, M, k):
    """The same batch through the host-pointer entry: H2D of the cost blocks, launch, D2H of all tables, into
    caller-owned buffers that are allocated (and touched) once, as a caller that runs frame after frame would.
    Twice: with plain (pageable) numpy buffers, and with the same buffers registered once with the engine
    (kbest_register_host_buffer: pinned + device-mapped, the result tables are written there by the kernel itself)."""
    from probabilisticsemslam_amd import engine as pk_engine
    B = costs.shape[0]
    costs = np.ascontiguousarray(costs)
    r4c = np.zeros((B, k, M), np.int32)
    c4r = np.zeros((B, k, N), np.int32)
    gain = np.zeros((B, k))
    nf = np.zeros(B, np.int32)
    o = eng._opts(False, None)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    legs = {}  # per leg: every call's time (two untimed calls first), so that the line carries median / min / max, not one shot

    def timed(with_c4r=True, leg=None, calls=11):
        ts = []
        for i in range(-2, calls):
            t0 = time.perf_counter()
            rc = eng.lib.kbest_batch_f64(eng.ctx, C.byref(o), B, N, M, None, None, p(costs), None, k, p(r4c), p(c4r) if with_c4r else None,
                                         p(gain), p(nf), None)
            dt = time.perf_counter() - t0
            assert rc == 0
            if i >= 0:
                ts.append(dt)
        if leg:
            legs[leg] = {"median_ms": 1e3 * float(np.median(ts)), "min_ms": 1e3 * min(ts), "max_ms": 1e3 * max(ts), "calls": len(ts)}
        return float(np.median(ts))

    pageable = timed(leg="pageable")
    ref = (r4c.copy(), c4r.copy(), gain.copy(), nf.copy())
    for a in (r4c, c4r, gain, nf):
        a[...] = 0
    registered = None
    r8, c8 = np.zeros((B, k, M), np.int8), np.zeros((B, k, N), np.int8)
    regs = []  # what is registered right now (unregistered in the finally block whatever happens)

    def register(*arrays):
        for a in arrays:
            eng.register_host(a)
            regs.append(a)

    def unregister(*arrays):
        for a in arrays:
            eng.unregister_host(a)
            regs[:] = [x for x in regs if x is not a]

    try:
        try:
            register(costs, r4c, c4r, gain, nf)
        except pk_engine.KBestError as ex:  # only a failed registration falls back to the pageable number
            err = repr(ex)
        else:
            registered = timed(leg="registered")
            # (a mismatch on any of the paths below is a failure of the bench, not a reason to fall back)
            assert all(np.array_equal(x, y) for x, y in zip(ref, (r4c, c4r, gain, nf))), "registered-buffer path differs from the copying path"
            no_c4r = timed(with_c4r=False, leg="registered_without_col4row")
            unregister(r4c, c4r)
            # the same tables as int8 (KBEST_FLAG_TABLES_I8): every index of a 64-row problem fits a byte
            register(r8, c8)
            o.flags |= pk_engine.KBEST_FLAG_TABLES_I8
            r4c_keep, c4r_keep = r4c, c4r
            r4c, c4r = r8, c8
            try:
                i8 = timed(leg="int8_tables")
                assert np.array_equal(r8, ref[0]) and np.array_equal(c8, ref[1]) and np.array_equal(gain, ref[2]), "int8 tables differ from the int32 tables"
                i8_no_c4r = timed(with_c4r=False, leg="int8_tables_without_col4row")
            finally:
                r4c, c4r = r4c_keep, c4r_keep
                o.flags &= ~pk_engine.KBEST_FLAG_TABLES_I8
    finally:
        for a in list(regs):
            try:
                eng.unregister_host(a)
            except pk_engine.KBestError:
                pass
        regs.clear()
    best = min(registered, pageable) if registered is not None else pageable
    # the same call with the int32 tables crossing the link as they are (round 3's path, KBEST_NO_NARROW: a second context)
    wide_ms = None
    try:
        import probabilisticsemslam_amd as pk
        os.environ["KBEST_NO_NARROW"] = "1"
        eng_w = pk.KBestEngine(0)
        del os.environ["KBEST_NO_NARROW"]
        eng_main, eng = eng, eng_w
        try:
            eng_w.register_host(costs, r4c, c4r, gain, nf)
            wide_ms = 1e3 * timed(leg="int32_tables_over_the_link", calls=5)
            assert all(np.array_equal(x, y) for x, y in zip(ref, (r4c, c4r, gain, nf))), "narrow staging differs from the int32 tables written by the kernel"
        finally:
            for a in (costs, r4c, c4r, gain, nf):
                try:
                    eng_w.unregister_host(a)
                except pk_engine.KBestError:
                    pass
            eng = eng_main
            eng_w.close()
    except pk_engine.KBestError:
        os.environ.pop("KBEST_NO_NARROW", None)
    out = {"value": float(nf.sum()) / best, "unit": "assignments/s", "ms": 1e3 * best,
           "ms_is": "the MEDIAN of 11 calls (after two untimed ones) of the better leg; every leg's median / min / max in `legs`",
           "legs": legs,
           "includes": "H2D of the cost blocks, kernel, D2H of row4col / col4row / gain / nf (host buffers in and out: kbest_batch_f64), into the "
                       "caller's int32 tables: the kernels write row4col as bytes into pinned staging, in four pieces; host threads of the "
                       "context widen a piece into row4col and its inverse col4row while the GPU works on the next one",
           "buffers": "caller-owned numpy arrays, reused across calls: the better (by median) of plain (pageable) arrays and of arrays registered once with "
                      "kbest_register_host_buffer (cost blocks then read in place by the kernel)",
           "registered_ms": None if registered is None else 1e3 * registered,
           "pageable_ms": 1e3 * pageable,
           "pageable_what": "the same call with unregistered (pageable) buffers: the cost blocks are uploaded piece by piece",
           "ms_int32_tables_over_the_link": wide_ms,
           "int32_over_the_link_what": "round 3's path (KBEST_NO_NARROW): the kernel writes the 107 MB of int32 tables into registered caller memory itself"}
    if registered is None:
        out["register_error"] = err
    else:
        out["ms_without_col4row"] = 1e3 * no_c4r
        out["without_col4row_what"] = ("col4row = NULL (legal: it is the inverse of row4col, and assignmentProb, the reference's caller, "
                                       "never reads it -- assignment.cpp:629): half the table bytes, the PCIe link no longer slows the kernel")
        out["ms_int8_tables"] = 1e3 * i8
        out["ms_int8_tables_without_col4row"] = 1e3 * i8_no_c4r
        out["int8_tables_what"] = ("KBEST_FLAG_TABLES_I8: row4col / col4row as int8 tables (same values; indices of <= 127-row problems fit a "
                                   "byte): a quarter of the table bytes cross PCIe; checked equal to the int32 tables")
        out["note"] = (f"{(r4c.nbytes + c4r.nbytes + gain.nbytes) / 1e6:.0f} MB of int32 tables leave the kernel over PCIe while it runs: "
                       "~35 GB/s is the link's rate for these stores, so the run cannot end before ~3.0 ms; see DESIGN.md section 6")
    return out


def multi_entry_dense(costs, N, M, k, G, ref_nf, ref_gain):
    """The same batch through the ONE-PROCESS multi-device entry (kbest_batch_f64_multi, kbest_multi.cpp) on G contexts of GPU 0
    ("logical devices"): host buffers in and out, one worker thread per device, the packed slices exchanged device to device.
    With G = 1 it is the single-device host path plus the exchange; with G = 8 the host side of config 4 as written (each
    device 1/8 of the batch) runs on one GPU, and the timeline shows what each device's thread did when."""
    import probabilisticsemslam_amd as pk
    from probabilisticsemslam_amd import engine as pk_engine
    multi = pk.KBestMulti([0] * G)
    B = costs.shape[0]
    costs = np.ascontiguousarray(costs)
    # caller-owned tables, allocated and touched once (as in host_inclusive_dense: a caller that runs batch after batch)
    r4c, c4r, gain, nf = np.zeros((B, k, M), np.int32), np.zeros((B, k, N), np.int32), np.zeros((B, k)), np.zeros(B, np.int32)
    o = pk_engine.KBestOpts()
    multi.lib.kbest_default_opts(C.byref(o))
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        rc = multi.lib.kbest_batch_f64_multi(multi.m, C.byref(o), B, N, M, None, None, p(costs), k, p(r4c), p(c4r), p(gain), p(nf))
        dt = time.perf_counter() - t0
        assert rc == 0, multi.lib.kbest_multi_last_error(multi.m)
        best = dt if best is None or dt < best else best
    out = (nf, r4c, c4r, gain)
    tl = multi.timeline()
    agree = bool(multi.tables_agree())
    xbytes, _ = multi.exchange_bytes()
    # the same matrices in SUBTREE mode (the north star's latency mode: every device enumerates its share of the root's subtrees of
    # every matrix; all-gather of the top-k costs + sum all-reduce of the winners' rows), a small batch: what the exchange moves
    sub = None
    try:
        Bs = 32
        r2, c2, g2, n2 = np.zeros((Bs, k, M), np.int32), np.zeros((Bs, k, N), np.int32), np.zeros((Bs, k)), np.zeros(Bs, np.int32)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = multi.lib.kbest_batch_f64_multi_ex(multi.m, C.byref(o), pk_engine.KBEST_MULTI_SUBTREE, 0, Bs, N, M, None, None, p(costs), k, p(r2), p(c2), p(g2), p(n2))
            ts.append(time.perf_counter() - t0)
            assert rc == 0, multi.lib.kbest_multi_last_error(multi.m)
        sb, path = multi.exchange_bytes()
        sub = {"matrices": Bs, "ms": 1e3 * min(ts), "bytes_inbound_per_device": sb, "path": {1: "gains first", 2: "whole lists"}.get(path, path),
               "equals_batch_mode": bool((n2 == ref_nf[:Bs]).all() and (g2.view(np.int64) == ref_gain[:Bs].view(np.int64)).all() and (r2 == r4c[:Bs]).all()),
               "whole_lists_would_move": (G - 1) * Bs * k * (8 + M) if G > 1 else 0}
    except Exception as ex:  # noqa: BLE001
        sub = {"error": repr(ex)}
    multi.close()
    assert (out[0] == ref_nf).all() and (out[3].view(np.int64) == ref_gain.view(np.int64)).all(), "multi-device entry differs from the single-device result"
    return {"devices": G, "ms": 1e3 * best, "value": float(out[0].sum()) / best, "unit": "assignments/s", "tables_agree": agree,
            "exchange_bytes_inbound_per_device": xbytes, "subtree_mode": sub,
            "timeline_ms": {"what": "host times of the last call per device, ms since entry: worker started, first upload issued, first "
                                    "kernel issued, fed (own results back), exchange issued, done (kbest_multi_timeline)",
                            "per_device": [[round(1e3 * float(x), 3) for x in row] for row in tl]},
            "no_device_waits_for_anothers_copy": bool(tl[:, 1].max() < tl[:, 3].min())}


def two_in_flight(torch, dev, cfg, B, ms_single, steps=20):
    """Steady state of a caller that streams batch after batch: two contexts (each its own workspace) on two streams, batches
    alternating, so that the next batch's workgroups take the CUs that the current batch's last generation leaves idle.
    Not the headline (a step there is one launch after the other on one stream); same tables."""
    import probabilisticsemslam_amd as pk
    from probabilisticsemslam_amd import workloads as wl
    _, N, M, k, seed = wl.DENSE_CONFIGS[cfg]
    d_cost = torch.from_numpy(wl.dense_batch(B, N, M, seed)).to(dev)
    engs = [pk.KBestEngine(dev.index or 0) for _ in range(2)]
    strs = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [(torch.empty((B, k, M), dtype=torch.int32, device=dev), torch.empty((B, k, N), dtype=torch.int32, device=dev),
             torch.empty((B, k), dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(2)]
    for e in engs:
        e.reserve(B, N, k)
    for i in range(2):
        engs[i].kbest_dev(d_cost, B, N, M, k, *outs[i], stream=strs[i].cuda_stream)
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        for i in range(steps):
            engs[i % 2].kbest_dev(d_cost, B, N, M, k, *outs[i % 2], stream=strs[i % 2].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None or dt < best else best
    same = all(torch.equal(outs[0][j], outs[1][j]) for j in range(4))
    found = int(outs[0][3].sum().item())
    for e in engs:
        e.close()
    return {"ms_per_batch": 1e3 * best, "value": found / best, "unit": "assignments/s", "steps": steps, "same_tables": bool(same),
            "speedup_vs_one_in_flight": ms_single / (1e3 * best),
            "what": "two kbest contexts on two HIP streams, batches alternating (each context has its own hypothesis workspace): "
                    "the last generation of one batch's workgroups leaves CUs idle (1 024 matrices on 512 resident slots end at ~2.6 "
                    "matrix lifetimes, not 2.0) and the other batch's workgroups take them; a launch by itself lasts longer, "
                    "batches complete faster.  Secondary number: the headline and its roofline are one launch at a time"}


def equal_costs_entry(eng, torch, dev, tstream, B=4, N=64, M=64, k=1000):
    """ADVICE r5's case: all-EQUAL costs at a large k -- one run of k exactly equal gains per matrix, which the finishing launch has
    to bring into the canonical order (row4col lexicographic).  Its price = the same launch with and without KBEST_FLAG_NO_TIE_CHECK."""
    d_cost = torch.zeros((B, N * M), dtype=torch.float64, device=dev)
    d_r = torch.empty((B, k, M), dtype=torch.int32, device=dev)
    d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
    d_g = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_n = torch.empty(B, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    eng.reserve(B, N, k)
    out = {}
    for name, tc in (("ms_with_the_canonical_order", True), ("ms_without", False)):
        ts = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=tstream.cuda_stream, d_tie_flags=d_f if tc else None, tie_check=tc)
            torch.cuda.synchronize()
            if i:
                ts.append(1e3 * (time.perf_counter() - t0))
        out[name] = float(np.median(ts))
        if tc:
            r = d_r.cpu().numpy()
            out["ordered"] = bool(all([tuple(x) for x in r[b]] == sorted(tuple(x) for x in r[b]) for b in range(B)))
            out["flags"] = [int(x) for x in d_f.cpu().numpy()]
    out["workload"] = f"{B} x {N}x{M} matrices of all-equal costs, k={k}: one run of k equal gains each (KBEST_TIE_INSIDE | KBEST_TIE_BOUNDARY)"
    out["what"] = "the finishing launch orders a long run by a radix sort over the columns (kbest_ties.h, round 6; L^2 lexicographic comparisons before)"
    return out


def reference_order_entry(eng, torch, dev, tstream):
    """KBEST_FLAG_REFERENCE_ORDER (kbest_exact.hip): the reference's own order of operations on the device -- what it costs next to
    the default kernels, on 256 of the headline's matrices and on 1 000 integer-cost 28x10 problems (masses of exact ties)."""
    from probabilisticsemslam_amd import workloads as wl
    out = {}
    rng = np.random.default_rng(3)
    for name, costs, N, M, k in (("dense_64x64_k200", wl.dense_batch(256, 64, 64, wl.DENSE_CONFIGS["c4"][4]), 64, 64, 200),
                                 ("integer_28x10_k200", rng.integers(0, 12, size=(1000, 280)).astype(np.float64), 28, 10, 200)):
        B = costs.shape[0]
        d_cost = torch.from_numpy(np.ascontiguousarray(costs)).to(dev)
        d_r = torch.empty((B, k, M), dtype=torch.int32, device=dev)
        d_c = torch.empty((B, k, N), dtype=torch.int32, device=dev)
        d_g = torch.empty((B, k), dtype=torch.float64, device=dev)
        d_n = torch.empty(B, dtype=torch.int32, device=dev)
        ms = {}
        for mode in (True, False):
            ts = []
            for i in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.kbest_dev(d_cost, B, N, M, k, d_r, d_c, d_g, d_n, stream=tstream.cuda_stream, reference_order=mode)
                torch.cuda.synchronize()
                if i:
                    ts.append(1e3 * (time.perf_counter() - t0))
            ms["reference_order" if mode else "default"] = min(ts)
            if mode:
                g_ref = d_g.cpu().numpy().copy()
                d_r_ref = d_r.cpu().numpy().copy()
        out[name] = {"problems": B, "ms_reference_order": ms["reference_order"], "ms_default": ms["default"],
                     "same_gains": bool((g_ref.view(np.int64) == d_g.cpu().numpy().view(np.int64)).all())}
        # the host entry: its default (the fast kernels + only the problems with an exact tie again on the reference-order kernel: the
        # reference's answer) next to the engine's own rule on ties (KBEST_FLAG_CANONICAL_TIES: tied levels completed in steps)
        hm = {}
        for key, kw in (("ms_host_canonical_ties", {"canonical_ties": True}), ("ms_host_default", {})):
            ts = []
            for i in range(3):
                t0 = time.perf_counter()
                res = eng.kbest(costs, N, M, k, tie_flags=True, **kw)
                if i:
                    ts.append(1e3 * (time.perf_counter() - t0))
            hm[key] = min(ts)
        out[name].update(hm)
        out[name]["problems_run_again"] = int((res[-1] & 8).astype(bool).sum())  # KBEST_TIE_REFERENCE
        out[name]["default_equals_reference_order"] = bool((res[3].view(np.int64) == g_ref.view(np.int64)).all() and
                                                                  (res[1] == d_r_ref).all())
    out["what"] = ("kbest_batch_f64_dev with KBEST_FLAG_REFERENCE_ORDER: the reference's algorithm as it stands, one to eight waves per problem, exact ties in "
                   "the reference's heap order (tests: bit-identical to the compiled reference's goldens incl. col4row) -- next to the default kernels; "
                   "ms_host_*: the host entry by default (only the problems with an exact tie among their k + 1 best gains run again on the "
                   "reference-order kernel: `problems_run_again`; the reference's answer) and with KBEST_FLAG_CANONICAL_TIES (the engine's own rule)")
    return out


def dense_entry(eng, torch, cfg, steps, warmup, dev, tstream, cpu_sample, no_cpu):
    """One BASELINE config as an entry of the `configs` block (single GPU)."""
    from probabilisticsemslam_amd import workloads as wl
    Bc = wl.DENSE_CONFIGS[cfg][0]
    m = run_dense(eng, torch, None, cfg, Bc, steps, warmup, 0, 1, dev, tstream, False, 0)
    N, M, k = m["N"], m["M"], m["k"]
    e = {"workload": f"{Bc} dense {N}x{M} cost matrices, k={k} (splitmix64 seed {m['seed']:#x}), kBest2D semantics",
         "steps": steps, "ms_per_step": 1e3 * m["dt"] / steps, "kernel_ms": m["kern_ms"],
         "value": m["found"] * steps / m["dt"], "unit": "assignments/s", "problems_per_s": Bc * steps / m["dt"],
         "parity_prune_vs_noprune": m["parity_self"], "mean_pushed_per_matrix": float(m["pushed"].mean()),
         "roofline": roofline_block(cfg, Bc, m["balg"], m["kern_ms"])}
    iss = issue_block(cfg)
    if iss:
        e["issue"] = iss
    if not no_cpu:
        cb, ref, p_cpu = cpu_dense(m["costs"], N, M, k, cpu_sample)
        tp = tables_parity(m, ref)
        cb["parity_vs_gpu"] = tp["all"]
        cb["parity_tables"] = tp
        e["cpu_baseline"] = cb
        e["speedup_vs_cpu_1core"] = e["value"] / cb["value"]
    return e


# ------------------------------------------------------------------------------------------------ config 5
def run_c5(eng, torch, steps, warmup, dev, tstream, no_cpu, F=1000, k=200, nL=20, nM=10, kernel_only=False):
    """C5: F streamed KITTI-like frames, (nL+nM) x nM raw cost blocks, through the fused association
    kernel (conditionCosts -> kBest2DCutoff(k, 42) -> weights -> scatter back).  Batched throughput with the blocks
    resident in HBM, the host-inclusive batched call, and the reference's own call pattern: one frame per call."""
    from probabilisticsemslam_amd import engine as pk_engine, workloads as wl
    import oracle_lib as ol
    frames = wl.kitti_like_frames(F, nL=nL, nM=nM)
    nR = nL + nM
    stream = tstream.cuda_stream
    raw = np.ascontiguousarray(np.concatenate(frames))
    h_nL = np.full(F, nL, np.int32)
    h_nM = np.full(F, nM, np.int32)
    h_nRow = np.full(F, nR, np.int32)
    h_coff = (np.arange(F, dtype=np.int64) * nR * nM)
    h_poff = (np.arange(F, dtype=np.int64) * nM * (nL + 1))
    d_cost = torch.from_numpy(raw).to(dev)
    d_nL, d_nM, d_nRow = (torch.from_numpy(a).to(dev) for a in (h_nL, h_nM, h_nRow))
    d_coff, d_poff = torch.from_numpy(h_coff).to(dev), torch.from_numpy(h_poff).to(dev)
    d_probs = torch.zeros(F * nM * (nL + 1), dtype=torch.float64, device=dev)
    d_nf = torch.zeros(F, dtype=torch.int32, device=dev)
    eng.reserve_assoc(F, nR, nM, k)
    torch.cuda.synchronize()

    def launch():
        eng.assoc_probs_dev(F, nR, nM, d_nL, d_nM, d_nRow, d_cost, d_coff, k, d_probs, d_poff, d_nf, stream=stream)

    for _ in range(warmup):
        launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        launch()
        ev[i][1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    nf = d_nf.cpu().numpy()
    probs = d_probs.cpu().numpy().reshape(F, nM, nL + 1)
    assert (nf >= 0).all(), "a frame did not fit the fused kernel"
    if kernel_only:  # under the profiler: only the timed launches
        return {"workload": f"{F} streamed frames (kernel only)", "frames": F, "steps": steps, "ms_per_step": 1e3 * dt / steps,
                "kernel_ms": kern_ms, "value": float(nf.sum()) * steps / dt, "unit": "assignments/s"}
    # -- the host-pointer entry: all frames in one call, and the reference's call pattern, one frame per call
    lib, ctx = eng.lib, eng.ctx
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    hp = np.zeros(F * nM * (nL + 1))
    hnf = np.zeros(F, np.int32)
    for _ in range(2):
        t1 = time.perf_counter()
        rc = lib.kbest_assoc_probs_batch_f64(ctx, F, p(h_nL), p(h_nM), p(raw), p(h_coff), k, p(hp), p(h_poff), p(hnf))
        host_ms = 1e3 * (time.perf_counter() - t1)
    assert rc == 0 and (hnf == nf).all() and np.array_equal(hp.reshape(probs.shape), probs)
    host_reg_ms = None
    try:  # the same call with the cost blocks and the probabilities registered once (read / written in place by the kernel)
        hp[:] = 0
        eng.register_host(raw, hp)
        for _ in range(3):
            t1 = time.perf_counter()
            rc = lib.kbest_assoc_probs_batch_f64(ctx, F, p(h_nL), p(h_nM), p(raw), p(h_coff), k, p(hp), p(h_poff), p(hnf))
            host_reg_ms = 1e3 * (time.perf_counter() - t1)
        eng.unregister_host(raw, hp)
        assert rc == 0 and np.array_equal(hp.reshape(probs.shape), probs)
    except pk_engine.KBestError:
        host_reg_ms = None
    one_l, one_m, zero = np.array([nL], np.int32), np.array([nM], np.int32), np.zeros(1, np.int64)
    op, onf = np.zeros(nM * (nL + 1)), np.zeros(1, np.int32)
    ncall = min(F, 400)
    lat = np.empty(ncall)
    # 300 untimed calls first: the HIP runtime grows its dispatch resources (signals, kernel-argument chunks) once, about
    # 200 launches into a process, and that one-off stall (35 ms measured) is not part of a frame's latency
    for i in range(-300, ncall):
        f = frames[i % F]
        t1 = time.perf_counter()
        lib.kbest_assoc_probs_batch_f64(ctx, 1, p(one_l), p(one_m), p(f), p(zero), k, p(op), p(zero), p(onf))
        if i >= 0:
            lat[i] = time.perf_counter() - t1
            if i < 8:
                assert np.array_equal(op.reshape(nM, nL + 1), probs[i])
    # -- the floor of a call: an EMPTY frame (no measurements: getAssignmentProbs returns at once, assignment.cpp:50-51) goes through
    #    exactly the same path -- pinned staging, one launch of the fused kernel (which returns at its shape test), the polled
    #    completion counter -- so its time is launch + completion and nothing else
    floor_us = None
    try:
        e_l, e_m = np.array([nL], np.int32), np.array([0], np.int32)
        ep, enf = np.zeros(1), np.zeros(1, np.int32)
        fl = np.empty(300)
        for i in range(-100, 300):
            t1 = time.perf_counter()
            rc = lib.kbest_assoc_probs_batch_f64(ctx, 1, p(e_l), p(e_m), p(frames[0]), p(zero), k, p(ep), p(zero), p(enf))
            if i >= 0:
                fl[i] = time.perf_counter() - t1
        if rc == 0:
            floor_us = {"us_mean": 1e6 * float(fl.mean()), "us_median": 1e6 * float(np.median(fl)), "calls": 300,
                        "what": "kbest_assoc_probs_batch_f64(B=1) on a frame with no measurements: the same zero-copy launch and "
                                "polled completion, no work in the kernel -- the part of a one-frame call that is not the kernel"}
    except Exception as ex:  # noqa: BLE001
        floor_us = {"error": repr(ex)}
    # -- the reference's REAL frame sizes ("3-5 measurements per frame", README.md:11): one frame per call, next to the
    #    reference's own conditionCosts + assignmentProb on one host core.  A GPU call cannot be shorter than its launch and
    #    completion (one_frame_per_call_floor); frames with this few assignments in all go through the exhaustive kernel
    #    (kbest_tiny.hip) instead of the enumeration (table: profiles/r04_crossover.json).
    small = []
    for (snL, snM) in ((6, 3), (6, 5)):
        sf = wl.kitti_like_frames(64, nL=snL, nM=snM, seed=0xC0FFEE + snL * 100 + snM)
        s_l, s_m = np.array([snL], np.int32), np.array([snM], np.int32)
        sp_, snf = np.zeros(snM * (snL + 1)), np.zeros(1, np.int32)
        slat = np.empty(200)
        for i in range(-100, 200):
            f = sf[i % 64]
            t1 = time.perf_counter()
            lib.kbest_assoc_probs_batch_f64(ctx, 1, p(s_l), p(s_m), p(f), p(zero), k, p(sp_), p(zero), p(snf))
            if i >= 0:
                slat[i] = time.perf_counter() - t1
        ent = {"nL": snL, "nM": snM, "us_mean": 1e6 * float(slat.mean()), "us_median": 1e6 * float(np.median(slat)), "calls": 200}
        if not no_cpu:
            kindS = "reference" if os.path.exists(ol.REF_ASSIGN_OFAST_SO) else "port"
            if kindS == "reference":  # (untimed first call: loads the checker's library)
                c0, r0 = ol.ref_condition_costs(sf[0], snL + snM, snM)
                ol.ref_assignment_prob(c0, len(r0) - snM, snM, k, ofast=True)
            t1 = time.perf_counter()
            for f in sf:
                if kindS == "reference":
                    c, ridx = ol.ref_condition_costs(f, snL + snM, snM)
                    ol.ref_assignment_prob(c, len(ridx) - snM, snM, k, ofast=True)
                else:
                    c, ridx = ol.condition_costs(f, snL + snM, snM)
                    ol.assignment_prob(c, len(ridx) - snM, snM, k)
            ent["cpu_us_per_frame"] = 1e6 * (time.perf_counter() - t1) / len(sf)
            ent["cpu_kind"] = kindS
            ent["speedup_vs_cpu_per_frame"] = ent["cpu_us_per_frame"] / ent["us_mean"]
        small.append(ent)
    # -- algorithmic bytes: raw block in, probabilities out, hypothesis states (P counted by the engine on the
    #    conditioned blocks in its no-prune mode, D = rows conditionCosts keeps)
    conds, idxs = eng.condition_costs(frames, [nR] * F, [nM] * F)
    D = np.array([len(i) for i in idxs], np.int32)
    coff = np.zeros(F, np.int64)
    coff[1:] = np.cumsum(D[:-1].astype(np.int64) * nM)
    _, _, _, _, pushed = eng.kbest(np.concatenate(conds), int(D.max()), nM, k, cutoff=42.0, nRow=D,
                                   nCol=np.full(F, nM, np.int32), costOff=coff, count_pushed=True, prune=False)
    balg = int(8 * nR * nM * F + 8 * nM * (nL + 1) * F + sum((int(pushed[i]) + int(nf[i]) - 1) * state_bytes(int(D[i])) for i in range(F)))
    out = {"workload": f"{F} streamed KITTI-like frames, raw ({nL}+{nM})x{nM} cost blocks (SURVEY 8(d) C5 generator, seed 0xc0ffee), "
                       f"conditionCosts -> kBest2DCutoff(k={k}, 42) -> weights -> scatter back, one fused launch",
           "frames": F, "steps": steps, "ms_per_step": 1e3 * dt / steps, "kernel_ms": kern_ms,
           "value": float(nf.sum()) * steps / dt, "unit": "assignments/s", "frames_per_s": F * steps / dt,
           "kernel_us_per_frame_batched": 1e3 * kern_ms / F,
           "host_inclusive_batched": {"ms": host_ms, "us_per_frame": 1e3 * host_ms / F,
                                      "includes": "kbest_assoc_probs_batch_f64 with all frames: copy into pinned staging memory, one launch that "
                                                  "reads / writes that memory in place, copy out",
                                      "ms_registered_buffers": host_reg_ms,
                                      "registered_what": "cost blocks and probabilities in memory registered once with "
                                                         "kbest_register_host_buffer: no staging copies either"},
           "one_frame_per_call": {"us_mean": 1e6 * float(lat.mean()), "us_median": 1e6 * float(np.median(lat)),
                                  "us_p95": 1e6 * float(np.percentile(lat, 95)), "us_max": 1e6 * float(lat.max()),
                                  "calls_over_1ms": int((lat > 1e-3).sum()), "slowest_calls": [int(i) for i in np.argsort(-lat)[:3]], "calls": ncall,
                                  "what": "kbest_assoc_probs_batch_f64(B=1) per frame, host buffers in and out (the reference's "
                                          "call pattern, system.cpp:268): zero-copy pinned staging, one launch, one stream sync"},
           "one_frame_per_call_floor": floor_us,
           "one_frame_per_call_small": small,
           "mean_rows_kept": float(D.mean()), "mean_pushed_per_frame": float(pushed.mean()),
           "roofline": roofline_block("c5", F, balg, kern_ms)}
    iss = issue_block("c5")
    if iss:
        out["issue"] = iss
    if not no_cpu:
        # the reference's own conditionCosts + assignmentProb (verbatim slices of assignment.cpp, -Ofast), one core
        kind = "reference" if os.path.exists(ol.REF_ASSIGN_OFAST_SO) else "port"
        ref_p = np.zeros_like(probs)
        tot = 0
        t1 = time.perf_counter()
        for i, f in enumerate(frames):
            if kind == "reference":
                c, ridx = ol.ref_condition_costs(f, nR, nM)
                q = ol.ref_assignment_prob(c, len(ridx) - nM, nM, k, ofast=True)
            else:
                c, ridx = ol.condition_costs(f, nR, nM)
                q, _ = ol.assignment_prob(c, len(ridx) - nM, nM, k)
            cl = len(ridx) - nM
            ref_p[i][:, np.asarray(ridx[:cl], dtype=np.int64)] = q[:, :cl]
            ref_p[i][:, nL] = q[:, cl]
        cpu_dt = time.perf_counter() - t1
        tot = int(nf.sum())
        err = float(np.abs(ref_p - probs).max())
        out["cpu_baseline"] = {"value": tot / cpu_dt, "unit": "assignments/s", "cores": 1, "kind": kind,
                               "us_per_frame": 1e6 * cpu_dt / F,
                               "sample": f"all {F} frames, conditionCosts + assignmentProb(k={k}) per frame as getAssignmentProbs "
                                         f"calls them (assignment.cpp:57-74), single thread, {cpu_dt:.2f} s",
                               "weights_max_abs_err_vs_gpu": err, "parity_vs_gpu": bool(err <= 1e-12)}
        out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        out["one_frame_per_call"]["speedup_vs_cpu_per_frame"] = out["cpu_baseline"]["us_per_frame"] / out["one_frame_per_call"]["us_mean"]
    return out


def visible_gpus():
    """Number of GPUs, learnt in a throw-away child so that the launcher itself never loads the HIP runtime."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launcher_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into: one rank per GPU under torch.distributed.run."""
    port = port or int(os.environ.get("MASTER_PORT", "29517"))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv, count=visible_gpus):
    """Spawn the ranks as a child process group, forward their output, return their exit code.  Fails fast (rc 2, one
    line on stderr, nothing spawned) when the node has fewer GPUs than asked for."""
    import subprocess
    have = count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but this node shows {have} GPU(s): nothing launched", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    p = subprocess.Popen(launcher_command(n_gpus, argv), env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    for ln in p.stdout:  # the ranks' stdout (rank 0's JSON line last) is this process' stdout
        sys.stdout.write(ln)
        sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5", "w128"])
    ap.add_argument("--batch", type=int, default=None, help="matrices per GPU (default: the config's B)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch PER GPU; strong: the config's batch in all (BASELINE configs[3] literally)")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `configs` block (c2, c3, c5 in the same run)")
    ap.add_argument("--no-host", action="store_true", help="skip value_host_inclusive (under the profiler: only the timed launches)")
    ap.add_argument("--kernel-only", action="store_true", help="c5 under the profiler: only the timed launches of the fused kernel")
    args = ap.parse_args()

    # (KBEST_BENCH_SELF_LAUNCH=1 takes the launcher path with one rank too: how it is exercised on a 1-GPU box)
    if (args.gpus > 1 or os.environ.get("KBEST_BENCH_SELF_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process becomes the launcher.  Nothing here has touched HIP (no torch
        # import yet), and the ranks are CHILD processes -- a process that has initialised the GPU is never re-exec'd.
        # (KBEST_BENCH_BACKEND=gloo deals the ranks to the GPUs there are -- two ranks on one GPU, the one-GPU test of this path)
        gloo = os.environ.get("KBEST_BENCH_BACKEND") == "gloo"
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], count=(lambda: args.gpus) if gloo else visible_gpus))

    import torch
    import torch.distributed as dist

    import probabilisticsemslam_amd as pk
    from probabilisticsemslam_amd import workloads as wl

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # KBEST_BENCH_BACKEND=gloo: the ranks' packed slices travel through the host (gloo) instead of RCCL, and the ranks are dealt to
    # the GPUs there are (LOCAL_RANK modulo the device count) -- RCCL refuses two ranks on one GPU, gloo does not: how EVERY line of
    # the world > 1 path (sharding, overlapped exchange, cross-rank checks, weak and strong scaling) runs on a one-GPU box
    backend = os.environ.get("KBEST_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"KBEST_BENCH_BACKEND={backend}: nccl or gloo")
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
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
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    eng = pk.KBestEngine(local)
    # a dedicated (non-null) HIP stream: the kernel, the timing events and the collective all go through it
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    assert tstream.cuda_stream != 0
    cpu_samples = {"c4": 512, "c3": 4096, "c2": 1024}  # (c3, c2: the whole batch -- the GPU result of every matrix is compared with the reference's)

    line = None
    if args.config == "c5":
        if world != 1:
            raise SystemExit("--config c5 is a single-GPU stream (frames are independent: run one stream per GPU)")
        e = run_c5(eng, torch, args.steps, args.warmup, dev, tstream, args.no_cpu, kernel_only=args.kernel_only)
        out = {"metric": "k-best assignments/sec (batched NxN cost matrices, k=200)", "value": e["value"], "unit": "assignments/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": e["ms_per_step"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": e["workload"], "frames_per_gpu": e["frames"], "k": 200, "parallelism": "single stream"}}
        out.update({kk: vv for kk, vv in e.items() if kk not in ("workload", "value", "unit", "steps", "ms_per_step", "frames")})
        line = json.dumps(out)
    else:
        Bc, N, M, k, seed = wl.DENSE_CONFIGS[args.config]
        if args.scaling == "strong":
            Btot = args.batch or Bc
            if Btot % world:
                raise SystemExit(f"strong scaling: {Btot} matrices do not divide over {world} ranks")
            B = Btot // world
        else:
            B = args.batch or Bc
        relays_before = eng.relay_launches()
        m = run_dense(eng, torch, dist if use_dist else None, args.config, B, args.steps, args.warmup, rank, world, dev, tstream,
                      use_dist, rank * B, backend=backend)
        relayed = eng.relay_launches() - relays_before  # (of this rank's launches: the no-prune run, the warm-ups, the timed steps)
        coll = m["collective"]
        t = torch.tensor([m["dt"], m["kern_ms"], (coll or {}).get("exposed_ms") or 0.0], dtype=torch.float64, device=dev)
        tot = torch.tensor([m["found"], m["balg"]], dtype=torch.float64, device=dev)
        if use_dist:
            all_reduce_(torch, dist, backend, t, dist.ReduceOp.MAX)
            all_reduce_(torch, dist, backend, tot, dist.ReduceOp.SUM)
        dt_max, kern_ms_max = float(t[0]), float(t[1])
        found_all, balg_all = float(tot[0]), float(tot[1])
        if coll and coll.get("exposed_ms") is not None:
            coll["exposed_ms_max_over_ranks"] = float(t[2])
        if rank == 0:
            what = "per GPU" if args.scaling == "weak" else f"in all, {B} per GPU"
            out = {
                "metric": "k-best assignments/sec (batched NxN cost matrices, k=200)",
                "value": found_all * args.steps / dt_max,
                "unit": "assignments/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt_max / args.steps,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"{B if args.scaling == 'weak' else B * world} dense {N}x{M} cost matrices {what}, k={k} "
                                       f"(BASELINE configs[3] shape, splitmix64 seed {seed:#x}), kBest2D semantics",
                           "matrices_per_gpu": B, "numRow": N, "numCol": M, "k": k, "parallelism": f"batch-sharded x{world}",
                           "collective": coll["what"] if use_dist else "none (one GPU)"},
                "problems_per_s": B * world * args.steps / dt_max,
                "kernel_ms": kern_ms_max,
                "parity_prune_vs_noprune": m["parity_self"],
                "launch": {"relay_launches": relayed, "of": (args.steps + args.warmup) * (2 if coll and coll.get("exposed_ms") is not None else 1),
                           "what": "launches of the 64-row kernel that ran as a relay: every matrix enumerated by three workgroups in turn, its LDS "
                                   "handed on through HBM (DESIGN.md section 2 point 12; KBEST_RELAY=0 launches it plainly)"},
                "roofline": roofline_block(args.config, B, balg_all / world, kern_ms_max,
                                           {"mean_pushed_per_matrix": float(m["pushed"].mean())}),
            }
            if coll:
                out["collective"] = coll
            iss = issue_block(args.config)
            if iss:
                out["issue"] = iss
            if world == 1 and not args.no_host:
                out["value_host_inclusive"] = host_inclusive_dense(eng, m["costs"], N, M, k)
            if world == 1 and not args.no_cpu:
                sample = args.cpu_sample or cpu_samples[args.config]
                cb, ref, p_cpu = cpu_dense(m["costs"], N, M, k, sample)
                ns = min(sample, B)
                tp = tables_parity(m, ref)
                cb["parity_vs_gpu"] = tp["all"]
                cb["parity_tables"] = tp
                if p_cpu is not None:
                    cb["pushed_matches_gpu"] = bool((p_cpu[:ns] == m["pushed"][:ns]).all())
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_1core"] = out["value"] / cb["value"]
                ca = cpu_all_cores(m["costs"], N, M, k)
                if ca is not None:
                    # the all-core run solves EVERY matrix of the batch with the reference: the tables of the timed launches (the
                    # relay path) are compared with all of them -- nf, gain bits, row4col, col4row
                    ca, ref_all = ca
                    tpa = tables_parity(m, ref_all)
                    ca["parity_vs_gpu"] = tpa["all"]
                    ca["parity_tables"] = tpa
                    cb["parity_vs_gpu"] = bool(cb["parity_vs_gpu"] and tpa["all"])
                    cb["parity_sample"] = f"all {tpa['matrices']} matrices of the timed launch (nf, gain bits, row4col, col4row) against the reference's tables from the all-core run"
                    out["cpu_baseline_all_cores"] = ca
                    out["speedup_vs_cpu_all_cores"] = out["value"] / ca["value"]
            if world == 1 and not args.no_extra and args.config == "c4" and args.batch is None:
                # the other BASELINE configs, measured in the same run (fewer steps each: the default run stays short)
                extra = {}
                for cfg in ("c2", "c3"):
                    extra[cfg] = dense_entry(eng, torch, cfg, 5, 1, dev, tstream, cpu_samples[cfg], args.no_cpu)
                extra["c5"] = run_c5(eng, torch, 5, 1, dev, tstream, args.no_cpu)
                try:  # (not a BASELINE config: the general-size kernel's profile case, 512 x 128x128 -- its roofline line)
                    extra["w128"] = dense_entry(eng, torch, "w128", 3, 1, dev, tstream, None, True)
                except Exception as ex:
                    extra["w128"] = {"error": repr(ex)}
                # BASELINE configs[3] AS WRITTEN -- 1 024 matrices over 8 GPUs -- gives each rank 128: measurable on one GPU.
                # Kernel alone, then the same with the RCCL path (one packed all-gather per step, world 1) inside the step.
                share = {}
                try:
                    ms = run_dense(eng, torch, None, "c4", Bc // 8, 10, 2, 0, 1, dev, tstream, False, 0)
                    share = {"workload": f"{Bc // 8} dense {N}x{M} matrices, k={k}: one rank's share of configs[3] over 8 GPUs",
                             "kernel_ms": ms["kern_ms"], "ms_per_step": 1e3 * ms["dt"] / 10, "value": ms["found"] * 10 / ms["dt"],
                             "unit": "assignments/s", "parity_prune_vs_noprune": ms["parity_self"]}
                    if not dist.is_initialized():
                        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                        os.environ.setdefault("MASTER_PORT", "29513")
                        os.environ.setdefault("RANK", "0")
                        os.environ.setdefault("WORLD_SIZE", "1")
                        dist.init_process_group("nccl", device_id=dev)
                        own_pg = True
                    else:
                        own_pg = False
                    md = run_dense(eng, torch, dist, "c4", Bc // 8, 10, 2, 0, 1, dev, tstream, True, 0)
                    share["with_rccl_allgather"] = {"ms_per_step": 1e3 * md["dt"] / 10, "kernel_ms": md["kern_ms"], "collective": md["collective"],
                                                    "what": "KBEST_BENCH_FORCE_DIST path: one packed all-gather per step on a 1-rank RCCL communicator, overlapped with the next step's kernel"}
                    # the headline's own batch with the exchange in the step (a 1-rank RCCL communicator: the call path, the packed int8
                    # slice, the overlap) -- what the exchange costs a step of the full-size workload
                    mf = run_dense(eng, torch, dist, "c4", Bc, 10, 2, 0, 1, dev, tstream, True, 0)
                    extra["c4_with_rccl_allgather"] = {"workload": f"{Bc} dense {N}x{M} matrices, k={k}: the headline batch, KBEST_BENCH_FORCE_DIST path",
                                                       "ms_per_step": 1e3 * mf["dt"] / 10, "kernel_ms": mf["kern_ms"], "collective": mf["collective"],
                                                       "vs_plain_step": (1e3 * mf["dt"] / 10) / out["ms_per_step"],
                                                       "parity_prune_vs_noprune": mf["parity_self"]}
                    if own_pg:
                        dist.destroy_process_group()
                    # strong scaling of configs[3] on 8 GPUs cannot beat (time of all 1 024 on one GPU) / (time of one share)
                    share["implied_8gpu_strong_scaling_ceiling"] = out["ms_per_step"] / share["with_rccl_allgather"]["ms_per_step"]
                    share["note"] = ("128 matrices leave half of the 256 CUs idle and every matrix alone on its CU: a matrix is bound by the "
                                     "latency of its own rounds (DESIGN.md section 8), so 8 GPUs are ~" +
                                     f"{share['implied_8gpu_strong_scaling_ceiling']:.1f}x one GPU on this config, not 8x; weak scaling (the default) is unaffected")
                except Exception as ex:  # never lose the headline over the extra entry
                    share["error"] = repr(ex)
                extra["c4_share8"] = share
                try:
                    extra["reference_order"] = reference_order_entry(eng, torch, dev, tstream)
                except Exception as ex:
                    extra["reference_order"] = {"error": repr(ex)}
                try:
                    extra["ties_all_equal_costs"] = equal_costs_entry(eng, torch, dev, tstream)
                except Exception as ex:
                    extra["ties_all_equal_costs"] = {"error": repr(ex)}
                try:
                    extra["c4_two_batches_in_flight"] = two_in_flight(torch, dev, "c4", Bc, out["ms_per_step"])
                except Exception as ex:
                    extra["c4_two_batches_in_flight"] = {"error": repr(ex)}
                # the one-process multi-device entry on logical devices of this GPU: 1 (against the single-device host path) and 8
                # (the host side of configs[3] as written)
                for G in (1, 8):
                    try:
                        me = multi_entry_dense(m["costs"], N, M, k, G, m["nf"], m["g"])
                        if G == 1 and "value_host_inclusive" in out:
                            me["vs_single_device_host_path_pageable"] = me["ms"] / out["value_host_inclusive"]["pageable_ms"]
                        extra[f"c4_multi_entry_{G}dev"] = me
                    except Exception as ex:
                        extra[f"c4_multi_entry_{G}dev"] = {"error": repr(ex)}
                out["configs"] = extra
            line = json.dumps(out)
    if use_dist:
        dist.destroy_process_group()
    # RCCL prints a banner through C stdio, which is flushed at exit, i.e. AFTER anything Python has printed: push
    # it out first so that the JSON line is the last line of stdout
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
