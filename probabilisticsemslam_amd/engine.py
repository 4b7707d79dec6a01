"""ctypes binding of include/kbest_c.h.  No compute happens in Python and there
is no fallback: if the HIP library is missing or no GPU is present, calls fail
loudly (KBestError)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KBEST_FLAG_NO_PRUNE = 1
KBEST_FLAG_COUNT_PUSHED = 2
KBEST_FLAG_TABLES_I8 = 64
KBEST_FLAG_NO_REORDER = 128
KBEST_FLAG_NO_TIE_CHECK = 512
KBEST_FLAG_NO_TIE_RESOLVE = 1024
KBEST_FLAG_REFERENCE_ORDER = 2048
KBEST_FLAG_REFERENCE_TIES = 4096
KBEST_FLAG_CANONICAL_TIES = 8192
# per-problem tie flags (kbest_c.h, "Order of exact ties")
KBEST_TIE_INSIDE, KBEST_TIE_BOUNDARY, KBEST_TIE_RESOLVED, KBEST_TIE_REFERENCE = 1, 2, 4, 8
KBEST_TIE_UNCHECKED, KBEST_TIE_UNORDERED, KBEST_TIE_UNRESOLVED = 1 << 28, 1 << 29, 1 << 30
KBEST_ROUTE_LANE, KBEST_ROUTE_SMALL, KBEST_ROUTE_FAST, KBEST_ROUTE_WIDE, KBEST_ROUTE_RELAY, KBEST_ROUTE_EXTRA = 1, 2, 4, 8, 16, 32
KBEST_ROUTE_EXACT = 64
KBEST_TIE_CAP = 4096
KBEST_MAX_DIM = 64        # rows handled by the LDS-resident kernel
KBEST_MAX_DIM_WIDE = 1024  # rows of the general-size kernel (beyond KBEST_MAX_DIM)
KBEST_MAX_DIM_EXACT = 16384  # rows handled at all (the reference-order kernel beyond KBEST_MAX_DIM_WIDE)

# every symbol include/kbest_c.h declares
C_ABI_SYMBOLS = (
    "kbest_default_opts", "kbest_create", "kbest_destroy", "kbest_strerror", "kbest_last_error",
    "kbest_device_count", "kbest_batch_f64_dev", "kbest_batch_f64", "kbest_reserve", "kbest_weights_batch_f64",
    "kbest_set_profile_buffer", "kbest_condition_costs_f64", "kbest_assoc_probs_batch_f64",
    "kbest_quadric_costs_f64", "kbest_quadric_assoc_probs_batch_f64", "kbest_bb_match_batch_f64",
    "kbest_bruteforce_probs_batch_f64", "kbest_assign_batch_f64", "kbest_to_probs_f64",
    "kbest_assoc_probs_batch_f64_dev", "kbest_reserve_assoc",
    "kbest_create_multi", "kbest_destroy_multi", "kbest_multi_size", "kbest_multi_last_error", "kbest_batch_f64_multi",
    "kbest_multi_tables_agree", "kbest_batch_f64_multi_ex", "kbest_merge_topk_f64_dev", "kbest_register_host_buffer",
    "kbest_unregister_host_buffer", "kbest_multi_timeline", "kbest_last_tie_flags", "kbest_set_assoc_tie_flags_dev",
    "kbest_relay_launches", "kbest_merge_topk_i8_f64_dev", "kbest_merge_gains_f64_dev", "kbest_multi_exchange_bytes",
    "kbest_last_route", "kbest_resolve_ties_dev", "kbest_multi_last_tie_flags", "kbest_reserve_exact",
    "kbest_set_reference_order",
)
KBEST_MULTI_STAMPS = 6
KBEST_MULTI_BATCH, KBEST_MULTI_SUBTREE = 0, 1


class KBestError(RuntimeError):
    pass


class KBestOpts(C.Structure):
    _fields_ = [("maximize", C.c_int32), ("use_cutoff", C.c_int32), ("cutoff", C.c_double), ("flags", C.c_uint32),
                ("root_col_offset", C.c_int32), ("root_col_stride", C.c_int32), ("tie_flags", C.c_void_p)]


def lib_path() -> str:
    # KBEST_LIB selects another in-tree build of the same library (the diagnostic libkbest_amd_prof.so)
    return os.path.join(_HERE, os.environ.get("KBEST_LIB", "libkbest_amd.so"))


_lib = None


def load_library():
    """Load the in-tree HIP library; raises KBestError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise KBestError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, i32p, i64p, dp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.kbest_default_opts.argtypes = [C.POINTER(KBestOpts)]
    lib.kbest_default_opts.restype = None
    lib.kbest_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.kbest_destroy.argtypes = [vp]
    lib.kbest_strerror.argtypes = [C.c_int]
    lib.kbest_strerror.restype = C.c_char_p
    lib.kbest_last_error.argtypes = [vp]
    lib.kbest_last_error.restype = C.c_char_p
    lib.kbest_device_count.restype = C.c_int
    lib.kbest_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    if hasattr(lib, "kbest_reserve_exact"):
        lib.kbest_reserve_exact.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.kbest_set_profile_buffer.argtypes = [vp, vp]
    lib.kbest_batch_f64_dev.argtypes = [vp, C.POINTER(KBestOpts), C.c_int, C.c_int, C.c_int, i32p, i32p, dp, i64p,
                                        C.c_int, i32p, i32p, dp, i32p, i64p, vp]
    lib.kbest_batch_f64.argtypes = [vp, C.POINTER(KBestOpts), C.c_int, C.c_int, C.c_int, i32p, i32p, dp, i64p,
                                    C.c_int, i32p, i32p, dp, i32p, i64p]
    lib.kbest_weights_batch_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, i64p, C.c_int, dp, i64p, i32p]
    lib.kbest_assoc_probs_batch_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, i64p, C.c_int, dp, i64p, i32p]
    lib.kbest_bruteforce_probs_batch_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, i64p, C.c_int, dp, i64p, i32p]
    lib.kbest_condition_costs_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, i64p, dp, i32p, i32p, C.c_int]
    lib.kbest_quadric_costs_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, dp, dp, dp, C.c_double, dp]
    lib.kbest_quadric_assoc_probs_batch_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, dp, dp, dp, C.c_double, C.c_int,
                                                        dp, i64p, i32p]
    lib.kbest_bb_match_batch_f64.argtypes = [vp, C.c_int, i32p, i32p, dp, dp, C.c_double, i32p]
    lib.kbest_assign_batch_f64.argtypes = [vp, C.c_int, C.c_int, C.c_int, i32p, i32p, dp, i64p, C.c_int, C.c_int, C.c_int,
                                           i32p, i32p, dp, dp, dp, i32p]
    lib.kbest_to_probs_f64.argtypes = [vp, dp, C.c_int64]
    lib.kbest_assoc_probs_batch_f64_dev.argtypes = [vp, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, dp, i64p, C.c_int, C.c_int,
                                                    dp, i64p, i32p, vp]
    lib.kbest_reserve_assoc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.kbest_create_multi.argtypes = [C.POINTER(vp), i32p, C.c_int]
    lib.kbest_destroy_multi.argtypes = [vp]
    lib.kbest_multi_size.argtypes = [vp]
    lib.kbest_multi_last_error.argtypes = [vp]
    lib.kbest_multi_last_error.restype = C.c_char_p
    lib.kbest_batch_f64_multi.argtypes = [vp, C.POINTER(KBestOpts), C.c_int, C.c_int, C.c_int, i32p, i32p, dp, C.c_int, i32p,
                                          i32p, dp, i32p]
    lib.kbest_multi_tables_agree.argtypes = [vp]
    lib.kbest_batch_f64_multi_ex.argtypes = [vp, C.POINTER(KBestOpts), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, dp,
                                             C.c_int, i32p, i32p, dp, i32p]
    lib.kbest_merge_topk_f64_dev.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int64, dp, i32p,
                                             i32p, vp]
    if hasattr(lib, "kbest_merge_gains_f64_dev"):
        lib.kbest_merge_topk_i8_f64_dev.argtypes = lib.kbest_merge_topk_f64_dev.argtypes
        lib.kbest_merge_gains_f64_dev.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, i32p, C.c_int, vp, dp, vp, i32p, i32p, vp]
        lib.kbest_multi_exchange_bytes.argtypes = [vp, C.POINTER(C.c_int)]
        lib.kbest_multi_exchange_bytes.restype = C.c_longlong
    if hasattr(lib, "kbest_multi_timeline"):  # (absent from older in-tree builds selected with KBEST_LIB for A/B runs)
        lib.kbest_multi_timeline.argtypes = [vp, dp, C.c_int]
    if hasattr(lib, "kbest_relay_launches"):
        lib.kbest_relay_launches.argtypes = [vp]
        lib.kbest_relay_launches.restype = C.c_longlong
    if hasattr(lib, "kbest_last_route"):
        lib.kbest_last_route.argtypes = [vp]
        lib.kbest_resolve_ties_dev.argtypes = [vp, C.POINTER(KBestOpts), C.c_int, C.c_int, C.c_int, i32p, i32p, dp, i64p, C.c_int, i32p, i32p, dp,
                                               i32p, vp]
        lib.kbest_multi_last_tie_flags.argtypes = [vp, i32p, C.c_int]
    if hasattr(lib, "kbest_last_tie_flags"):
        lib.kbest_last_tie_flags.argtypes = [vp, i32p, C.c_int]
        lib.kbest_set_assoc_tie_flags_dev.argtypes = [vp, vp]
    lib.kbest_register_host_buffer.argtypes = [vp, vp, C.c_size_t]
    lib.kbest_unregister_host_buffer.argtypes = [vp, vp]
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class KBestEngine:
    """One engine context = one GPU, one stream, one hypothesis-state workspace."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.kbest_create(C.byref(self.ctx), device)
        if rc != 0:
            raise KBestError(f"kbest_create(device={device}): {self.lib.kbest_strerror(rc).decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.kbest_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise KBestError(f"{self.lib.kbest_strerror(rc).decode()}: {self.lib.kbest_last_error(self.ctx).decode()}")

    def _opts(self, maximize, cutoff, flags=0, root_shard=None):
        o = KBestOpts()
        self.lib.kbest_default_opts(C.byref(o))
        o.maximize = int(bool(maximize))
        o.use_cutoff = int(cutoff is not None)
        o.cutoff = float(cutoff) if cutoff is not None else 0.0
        o.flags = flags
        if root_shard is not None:
            o.root_col_offset, o.root_col_stride = root_shard
        return o

    # ---- host buffers -----------------------------------------------------------------
    def kbest(self, costs, N, M, k, maximize=False, cutoff=None, nRow=None, nCol=None, costOff=None,
              count_pushed=False, prune=True, root_shard=None, tables_i8=False, reorder=True, tie_flags=False,
              tie_check=True, tie_resolve=True, reference_order=False, reference_ties=False, canonical_ties=False):
        """Batched kBest2D / kBest2DCutoff.  costs: (B, N*M) for uniform shapes, or a flat packed
        array with per-problem nRow/nCol/costOff (N, M are then the maxima).
        Returns (nf[B], row4col[B,k,M], col4row[B,k,N], gain[B,k]) (+ pushed[B] if count_pushed).
        tables_i8: the two tables come back as int8 (KBEST_FLAG_TABLES_I8; N <= 127).
        tie_flags: also return the per-problem KBEST_TIE_* flags (last element of the tuple).
        reference_order: KBEST_FLAG_REFERENCE_ORDER -- the reference's own order of operations (exact ties in its heap's order,
        col4row on padded columns as the reference names them; slow).
        reference_ties: KBEST_FLAG_REFERENCE_TIES -- the fast kernels, and every problem with an exact tie among its k + 1 best gains
        again on the reference-order kernel (KBEST_TIE_REFERENCE): the reference's answer everywhere, fast where nothing ties.  This
        is the DEFAULT of the synchronous entries (the flag is accepted and changes nothing).
        canonical_ties: KBEST_FLAG_CANONICAL_TIES -- the engine's own rule on exact ties instead ((gain, row4col) lexicographic; a
        level that straddles slot k completed in steps of up to 4 096 solutions)."""
        costs = np.ascontiguousarray(costs, dtype=np.float64)
        if nRow is None:
            costs = costs.reshape(-1, N * M)
            B = costs.shape[0]
        else:
            nRow = np.ascontiguousarray(nRow, dtype=np.int32)
            nCol = np.ascontiguousarray(nCol, dtype=np.int32)
            costOff = np.ascontiguousarray(costOff, dtype=np.int64)
            B = len(nRow)
        tdt = np.int8 if tables_i8 else np.int32
        r4c = np.empty((B, k, M), tdt)
        c4r = np.empty((B, k, N), tdt)
        gain = np.empty((B, k), np.float64)
        nf = np.empty(B, np.int32)
        pushed = np.zeros(B, np.int64) if count_pushed else None
        flags = ((KBEST_FLAG_COUNT_PUSHED if count_pushed else 0) | (0 if prune else KBEST_FLAG_NO_PRUNE) |
                 (KBEST_FLAG_TABLES_I8 if tables_i8 else 0) | (0 if reorder else KBEST_FLAG_NO_REORDER) |
                 (0 if tie_check else KBEST_FLAG_NO_TIE_CHECK) | (0 if tie_resolve else KBEST_FLAG_NO_TIE_RESOLVE) |
                 (KBEST_FLAG_REFERENCE_ORDER if reference_order else 0) | (KBEST_FLAG_REFERENCE_TIES if reference_ties else 0) |
                 (KBEST_FLAG_CANONICAL_TIES if canonical_ties else 0))
        o = self._opts(maximize, cutoff, flags, root_shard)
        tf = np.zeros(B, np.int32) if tie_flags else None
        if tf is not None:
            o.tie_flags = tf.ctypes.data
        self._check(self.lib.kbest_batch_f64(self.ctx, C.byref(o), B, N, M, _ptr(nRow), _ptr(nCol), _ptr(costs),
                                             _ptr(costOff), k, _ptr(r4c), _ptr(c4r), _ptr(gain), _ptr(nf),
                                             _ptr(pushed)))
        out = (nf, r4c, c4r, gain) + ((pushed,) if count_pushed else ()) + ((tf,) if tie_flags else ())
        return out

    def assign(self, costs, N, M, maximize=False, shift=True, gain_cols=0):
        """Batched assign2D (shift=True) / shortestPathCPP (shift=False) on uniform N x M problems, costs (B, N*M).
        Returns (feasible[B], row4col[B,M], col4row[B,N] (-1 = unassigned), gain[B], u[B,M], v[B,N])."""
        costs = np.ascontiguousarray(costs, dtype=np.float64).reshape(-1, N * M)
        B = costs.shape[0]
        r4c = np.empty((B, M), np.int32); c4r = np.empty((B, N), np.int32)
        g = np.empty(B); u = np.empty((B, M)); v = np.empty((B, N)); ok = np.empty(B, np.int32)
        self._check(self.lib.kbest_assign_batch_f64(self.ctx, B, N, M, None, None, _ptr(costs), None, int(bool(maximize)),
                                                    int(bool(shift)), int(gain_cols), _ptr(r4c), _ptr(c4r), _ptr(g), _ptr(u),
                                                    _ptr(v), _ptr(ok)))
        return ok, r4c, c4r, g, u, v

    def to_probs(self, x):
        """toProbs (assignment.h:19): returns exp(min - x) with the 42 gate."""
        x = np.array(x, dtype=np.float64).reshape(-1)
        self._check(self.lib.kbest_to_probs_f64(self.ctx, _ptr(x), x.size))
        return x

    def condition_costs(self, costs, nRows, nCols):
        """Batched conditionCosts.  Returns (list of conditioned 1-D blocks, list of rowIdx arrays)."""
        nRows = np.ascontiguousarray(nRows, dtype=np.int32)
        nCols = np.ascontiguousarray(nCols, dtype=np.int32)
        B = len(nRows)
        sizes = nRows.astype(np.int64) * nCols
        off = np.zeros(B, np.int64)
        off[1:] = np.cumsum(sizes)[:-1]
        flat = np.concatenate([np.ascontiguousarray(c, dtype=np.float64).reshape(-1) for c in costs])
        out = np.zeros_like(flat)
        good = np.zeros(B, np.int32)
        maxRow = int(nRows.max())
        ridx = np.zeros((B, maxRow), np.int32)
        self._check(self.lib.kbest_condition_costs_f64(self.ctx, B, _ptr(nRows), _ptr(nCols), _ptr(flat), _ptr(off),
                                                       _ptr(out), _ptr(good), _ptr(ridx), maxRow))
        return ([out[off[b]: off[b] + int(good[b]) * int(nCols[b])].copy() for b in range(B)],
                [ridx[b, : good[b]].copy() for b in range(B)])

    def relay_launches(self):
        """Launches of the 64-row kernel this context has made as a relay (diagnostic, kbest_relay_launches)."""
        return int(self.lib.kbest_relay_launches(self.ctx))

    def set_reference_order(self, on=True):
        """kbest_set_reference_order: the host-buffer association entries enumerate in the reference's own order of operations
        (on = 1 / True), or only their frames with a tie at slot k do (on = 2: the same answer, fused kernels wherever nothing ties)."""
        self.lib.kbest_set_reference_order.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.kbest_set_reference_order(self.ctx, int(on)))

    def last_route(self):
        """KBEST_ROUTE_* bits of the kernel(s) this context's last k-best launch went to (diagnostic, kbest_last_route)."""
        return int(self.lib.kbest_last_route(self.ctx))

    def last_tie_flags(self):
        """KBEST_TIE_* flags of the problems of this context's last synchronous call (kbest_last_tie_flags)."""
        n = self.lib.kbest_last_tie_flags(self.ctx, None, 0)
        out = np.zeros(max(n, 0), np.int32)
        if n > 0:
            self.lib.kbest_last_tie_flags(self.ctx, _ptr(out), n)
        return out

    def set_assoc_tie_flags_dev(self, d_flags):
        """Where assoc_probs_dev writes its frames' KBEST_TIE_* flags (an int32 [B] torch CUDA tensor, or None)."""
        self._check(self.lib.kbest_set_assoc_tie_flags_dev(self.ctx, None if d_flags is None else C.c_void_p(d_flags.data_ptr())))

    def weights(self, costs, nL, nM, k, condition=False, brute_force=False):
        """Batched assignmentProb (condition=False) or, with condition=True, the whole
        conditionCosts -> assignmentProb -> scatter chain of getAssignmentProbs on raw cost blocks.
        costs: list of 1-D column-major (nL+nM) x nM blocks.  Returns (list of [nM, nL+1] arrays, nf[B])."""
        nL = np.ascontiguousarray(nL, dtype=np.int32)
        nM = np.ascontiguousarray(nM, dtype=np.int32)
        B = len(nL)
        sizes = [(int(nL[b]) + int(nM[b])) * int(nM[b]) for b in range(B)]
        psizes = [int(nM[b]) * (int(nL[b]) + 1) for b in range(B)]
        costOff = np.zeros(B, np.int64)
        probOff = np.zeros(B, np.int64)
        costOff[1:] = np.cumsum(sizes)[:-1]
        probOff[1:] = np.cumsum(psizes)[:-1]
        flat = np.concatenate([np.ascontiguousarray(c, dtype=np.float64).reshape(-1) for c in costs])
        probs = np.zeros(int(sum(psizes)), np.float64)
        nf = np.zeros(B, np.int32)
        fn = (self.lib.kbest_bruteforce_probs_batch_f64 if brute_force else
              self.lib.kbest_assoc_probs_batch_f64 if condition else self.lib.kbest_weights_batch_f64)
        self._check(fn(self.ctx, B, _ptr(nL), _ptr(nM), _ptr(flat), _ptr(costOff), k, _ptr(probs), _ptr(probOff),
                       _ptr(nf)))
        out = [probs[probOff[b]: probOff[b] + psizes[b]].reshape(int(nM[b]), int(nL[b]) + 1) for b in range(B)]
        return out, nf

    @staticmethod
    def _pack_quadrics(frames):
        """frames: list of (landMean (nL,3), landCov (nL,3,3), measMean (nM,3), measCov (nM,3,3))."""
        nL = np.array([len(f[0]) for f in frames], np.int32)
        nM = np.array([len(f[2]) for f in frames], np.int32)
        cat = lambda i, w: np.ascontiguousarray(np.concatenate([np.asarray(f[i], np.float64).reshape(-1, w) for f in frames]))  # noqa: E731
        return nL, nM, cat(0, 3), cat(1, 9), cat(2, 3), cat(3, 9)

    def quadric_costs(self, frames, gate):
        """Batched computeQuadricCostMatrix.  Returns a list of (nL+nM)*nM column-major blocks."""
        nL, nM, lm, lc, mm, mc = self._pack_quadrics(frames)
        sizes = (nL.astype(np.int64) + nM) * nM
        out = np.zeros(int(sizes.sum()))
        self._check(self.lib.kbest_quadric_costs_f64(self.ctx, len(frames), _ptr(nL), _ptr(nM), _ptr(lm), _ptr(lc), _ptr(mm),
                                                     _ptr(mc), float(gate), _ptr(out)))
        off = np.concatenate([[0], np.cumsum(sizes)])
        return [out[off[b]: off[b + 1]] for b in range(len(frames))]

    def quadric_assoc_probs(self, frames, gate, k):
        """getAssignmentProbs from (mean, covariance) pairs: list of [nM, nL+1] arrays, nf."""
        nL, nM, lm, lc, mm, mc = self._pack_quadrics(frames)
        psizes = nM.astype(np.int64) * (nL + 1)
        poff = np.zeros(len(frames), np.int64)
        poff[1:] = np.cumsum(psizes)[:-1]
        probs = np.zeros(int(psizes.sum()))
        nf = np.zeros(len(frames), np.int32)
        self._check(self.lib.kbest_quadric_assoc_probs_batch_f64(self.ctx, len(frames), _ptr(nL), _ptr(nM), _ptr(lm), _ptr(lc),
                                                                 _ptr(mm), _ptr(mc), float(gate), k, _ptr(probs), _ptr(poff),
                                                                 _ptr(nf)))
        return [probs[poff[b]: poff[b] + psizes[b]].reshape(int(nM[b]), int(nL[b]) + 1) for b in range(len(frames))], nf

    def bb_match(self, boxesL, boxesR, gate):
        """Batched asgnBB.  boxesL / boxesR: lists of (n, 5) arrays (xmin, ymin, xmax, ymax, xOffset)."""
        nL = np.array([len(b) for b in boxesL], np.int32)
        nR = np.array([len(b) for b in boxesR], np.int32)
        bl = np.ascontiguousarray(np.concatenate([np.asarray(b, np.float64).reshape(-1, 5) for b in boxesL]))
        br = np.ascontiguousarray(np.concatenate([np.asarray(b, np.float64).reshape(-1, 5) for b in boxesR] + [np.zeros((0, 5))]))
        asg = np.full(int(nL.sum()), -9, np.int32)
        self._check(self.lib.kbest_bb_match_batch_f64(self.ctx, len(boxesL), _ptr(nL), _ptr(nR), _ptr(bl), _ptr(br),
                                                      float(gate), _ptr(asg)))
        off = np.concatenate([[0], np.cumsum(nL)])
        return [asg[off[b]: off[b + 1]] for b in range(len(boxesL))]

    # ---- device buffers (torch tensors already resident in HBM) -------------------------
    def register_host(self, *arrays):
        """kbest_register_host_buffer on numpy arrays the caller keeps using (cost blocks, result tables): pinned and mapped,
        so that kbest_batch_f64 moves them without staging copies.  The arrays must stay alive until unregister_host."""
        for a in arrays:
            self._check(self.lib.kbest_register_host_buffer(self.ctx, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def unregister_host(self, *arrays):
        for a in arrays:
            self._check(self.lib.kbest_unregister_host_buffer(self.ctx, a.ctypes.data_as(C.c_void_p)))

    def reserve(self, B, N, k):
        self._check(self.lib.kbest_reserve(self.ctx, B, N, k))

    def kbest_dev(self, d_cost, B, N, M, k, d_row4col, d_col4row, d_gain, d_nf, maximize=False, cutoff=None,
                  d_pushed=None, prune=True, stream=None, root_shard=None, d_nRow=None, d_nCol=None, d_costOff=None,
                  tables_i8=False, d_tie_flags=None, tie_check=True, reference_order=False):
        """Asynchronous launch on `stream` (a raw hipStream_t integer, e.g.
        torch.cuda.current_stream().cuda_stream).  All d_* are torch CUDA tensors (tables_i8: d_row4col / d_col4row int8;
        d_tie_flags: int32 [B], receives the KBEST_TIE_* flags)."""
        flags = ((KBEST_FLAG_COUNT_PUSHED if d_pushed is not None else 0) | (0 if prune else KBEST_FLAG_NO_PRUNE) |
                 (KBEST_FLAG_TABLES_I8 if tables_i8 else 0) | (0 if tie_check else KBEST_FLAG_NO_TIE_CHECK) |
                 (KBEST_FLAG_REFERENCE_ORDER if reference_order else 0))
        o = self._opts(maximize, cutoff, flags, root_shard)
        if d_tie_flags is not None:
            o.tie_flags = d_tie_flags.data_ptr()
        # the C entry never allocates (kbest_c.h): size the workspace here (a no-op once it is large enough)
        if reference_order or N > KBEST_MAX_DIM_WIDE:
            self._check(self.lib.kbest_reserve_exact(self.ctx, B, N, M, k))
        else:
            self.reserve(B, N, k)

        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr())

        self._check(self.lib.kbest_batch_f64_dev(self.ctx, C.byref(o), B, N, M, dp(d_nRow), dp(d_nCol), dp(d_cost),
                                                 dp(d_costOff), k, dp(d_row4col), dp(d_col4row), dp(d_gain), dp(d_nf),
                                                 dp(d_pushed), C.c_void_p(stream) if stream else None))


    def resolve_ties_dev(self, d_cost, B, N, M, k, d_row4col, d_col4row, d_gain, d_tie_flags, maximize=False, cutoff=None, stream=None,
                         d_nRow=None, d_nCol=None, d_costOff=None, tables_i8=False, reference_ties=False, canonical_ties=False):
        """kbest_resolve_ties_dev: the synchronous second call behind kbest_dev -- completes the gain levels that straddle slot k
        in the device tables (same arguments as the launch).  By default (reference_ties: the accepted no-op flag) every problem flagged
        with a tie is replaced by the reference-order kernel's tables; canonical_ties: KBEST_FLAG_CANONICAL_TIES -- the engine's own
        rule instead (levels that straddle slot k completed in steps)."""
        o = self._opts(maximize, cutoff, (KBEST_FLAG_TABLES_I8 if tables_i8 else 0) | (KBEST_FLAG_REFERENCE_TIES if reference_ties else 0) |
                       (KBEST_FLAG_CANONICAL_TIES if canonical_ties else 0))

        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.kbest_resolve_ties_dev(self.ctx, C.byref(o), B, N, M, dp(d_nRow), dp(d_nCol), dp(d_cost), dp(d_costOff), k,
                                                    dp(d_row4col), dp(d_col4row), dp(d_gain), dp(d_tie_flags),
                                                    C.c_void_p(stream) if stream else None))

    def merge_topk_dev(self, B, n_shard, k, M, d_gain, d_row4col, d_nf, shard_stride_bytes, d_out_gain, d_out_row4col, d_out_nf,
                       maximize=False, stream=None, tables_i8=False):
        """kbest_merge_topk_f64_dev (tables_i8: kbest_merge_topk_i8_f64_dev -- the shards' row4col tables are int8): k-way merge
        of per-shard k-best lists (torch tensors / device pointers)."""
        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        fn = self.lib.kbest_merge_topk_i8_f64_dev if tables_i8 else self.lib.kbest_merge_topk_f64_dev
        self._check(fn(self.ctx, B, n_shard, k, M, int(bool(maximize)), dp(d_gain), dp(d_row4col), dp(d_nf), int(shard_stride_bytes),
                       dp(d_out_gain), dp(d_out_row4col), dp(d_out_nf), C.c_void_p(stream) if stream else None))

    def merge_gains_dev(self, B, n_shard, k, M, d_gain, d_nf, own_shard, d_own_row4col8, d_out_gain, d_out_row4col8, d_out_nf, d_tied,
                        maximize=False, stream=None):
        """kbest_merge_gains_f64_dev: the global k-best heap from all shards' gains [S,B,k] / nf [S,B] and this rank's own
        rows (int8 [B,k,M]); d_out_row4col8 (zeroed by the caller) receives the own winners' rows, d_tied (zeroed) the tie word."""
        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.kbest_merge_gains_f64_dev(self.ctx, B, n_shard, k, M, int(bool(maximize)), dp(d_gain), dp(d_nf), int(own_shard),
                                                       dp(d_own_row4col8), dp(d_out_gain), dp(d_out_row4col8), dp(d_out_nf), dp(d_tied),
                                                       C.c_void_p(stream) if stream else None))

    def reserve_assoc(self, B, maxRawRow, maxCol, k):
        self._check(self.lib.kbest_reserve_assoc(self.ctx, B, maxRawRow, maxCol, k))

    def assoc_probs_dev(self, B, maxRawRow, maxCol, d_nL, d_nM, d_nRow, d_cost, d_costOff, k, d_probs, d_probOff, d_nf,
                        condition=True, stream=None):
        """Fused association on device buffers (torch CUDA tensors), asynchronous on `stream`: one launch."""
        def dp(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.lib.kbest_assoc_probs_batch_f64_dev(self.ctx, B, maxRawRow, maxCol, dp(d_nL), dp(d_nM), dp(d_nRow),
                                                             dp(d_cost), dp(d_costOff), k, int(bool(condition)), dp(d_probs),
                                                             dp(d_probOff), dp(d_nf), C.c_void_p(stream) if stream else None))


class KBestMulti:
    """Multi-device engine of include/kbest_c.h: one context per GPU in ONE process, contiguous block sharding, RCCL
    all-gather of the packed result tables (kbest_multi.cpp)."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        self.m = C.c_void_p()
        rc = self.lib.kbest_create_multi(C.byref(self.m), _ptr(ids), len(ids))
        if rc != 0:
            raise KBestError(f"kbest_create_multi({list(ids)}): {self.lib.kbest_strerror(rc).decode()}")

    def close(self):
        if getattr(self, "m", None):
            self.lib.kbest_destroy_multi(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kbest(self, costs, N, M, k, maximize=False, cutoff=None, nRow=None, nCol=None, subtree=False, n_shard=0, reference_ties=False,
              canonical_ties=False):
        """Batch mode (default): contiguous blocks of the batch per device.  subtree=True: every device enumerates its share
        of the root's subtrees of every matrix (n_shard shards, 0 = one per device), one all-gather of the packed lists,
        k-way merge on the device (kbest_merge.hip)."""
        costs = np.ascontiguousarray(costs, dtype=np.float64).reshape(-1, N * M)
        B = costs.shape[0]
        r4c = np.empty((B, k, M), np.int32); c4r = np.empty((B, k, N), np.int32)
        gain = np.empty((B, k)); nf = np.empty(B, np.int32)
        o = KBestOpts()
        self.lib.kbest_default_opts(C.byref(o))
        o.maximize = int(bool(maximize)); o.use_cutoff = int(cutoff is not None); o.cutoff = float(cutoff or 0.0)
        if reference_ties:  # (batch mode: kbest_c.h, KBEST_FLAG_REFERENCE_TIES -- the default; accepted)
            o.flags |= KBEST_FLAG_REFERENCE_TIES
        if canonical_ties:  # (the engine's own rule on exact ties)
            o.flags |= KBEST_FLAG_CANONICAL_TIES
        if nRow is not None:
            nRow = np.ascontiguousarray(nRow, dtype=np.int32); nCol = np.ascontiguousarray(nCol, dtype=np.int32)
        if subtree:
            rc = self.lib.kbest_batch_f64_multi_ex(self.m, C.byref(o), KBEST_MULTI_SUBTREE, int(n_shard), B, N, M, _ptr(nRow),
                                                   _ptr(nCol), _ptr(costs), k, _ptr(r4c), _ptr(c4r), _ptr(gain), _ptr(nf))
        else:
            rc = self.lib.kbest_batch_f64_multi(self.m, C.byref(o), B, N, M, _ptr(nRow), _ptr(nCol), _ptr(costs), k, _ptr(r4c),
                                                _ptr(c4r), _ptr(gain), _ptr(nf))
        if rc != 0:
            raise KBestError(f"{self.lib.kbest_strerror(rc).decode()}: {self.lib.kbest_multi_last_error(self.m).decode()}")
        return nf, r4c, c4r, gain

    def tables_agree(self):
        return self.lib.kbest_multi_tables_agree(self.m) == 1

    def last_tie_flags(self):
        """KBEST_TIE_* flags of the problems of the last batch-mode call (kbest_multi_last_tie_flags)."""
        n = self.lib.kbest_multi_last_tie_flags(self.m, None, 0)
        out = np.zeros(max(n, 0), np.int32)
        if n > 0:
            self.lib.kbest_multi_last_tie_flags(self.m, _ptr(out), n)
        return out

    def exchange_bytes(self):
        """(bytes that arrived at one device in the exchanges of the last call, path: 0 batch, 1 subtree gains first, 2 subtree whole lists)."""
        path = C.c_int(0)
        n = self.lib.kbest_multi_exchange_bytes(self.m, C.byref(path))
        return int(n), int(path.value)

    def timeline(self):
        """Host times of the last call per device: array [nDev, 6] of seconds since the call was entered (kbest_multi_timeline)."""
        n = self.lib.kbest_multi_size(self.m)
        out = np.zeros((n, KBEST_MULTI_STAMPS))
        self.lib.kbest_multi_timeline(self.m, _ptr(out), n)
        return out


# ---- reference-named conveniences (B = 1), mirroring shortestPathCPP.hpp / assignment.h -------------
_default = None


def _engine():
    global _default
    if _default is None:
        _default = KBestEngine(0)
    return _default


def kBest2D(k, numRow, numCol, maximize, C_):
    """shortestPathCPP.hpp:204-212.  Returns (numFound, col4rowBest[k,numRow], row4colBest[k,numCol], gainBest[k])."""
    nf, r4c, c4r, g = _engine().kbest(np.asarray(C_).reshape(1, -1), numRow, numCol, k, maximize)
    return int(nf[0]), c4r[0], r4c[0], g[0]


def kBest2DCutoff(k, numRow, numCol, maximize, C_, cutoff):
    """shortestPathCPP.hpp:256-265."""
    nf, r4c, c4r, g = _engine().kbest(np.asarray(C_).reshape(1, -1), numRow, numCol, k, maximize, cutoff)
    return int(nf[0]), c4r[0], r4c[0], g[0]


def assignmentProb(costMatrix, nL, nM, k):
    """assignment.h:11.  Returns probs[nM][nL+1]."""
    out, _ = _engine().weights([costMatrix], [nL], [nM], k)
    return out[0]
