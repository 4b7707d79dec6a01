"""ctypes access to the CPU checker (oracle/) for tests, smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libkbest_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_kbest.so")
REF_OFAST_SO = os.path.join(ORACLE_DIR, "_ref", "libref_kbest_ofast.so")
REF_ASSIGN_SO = os.path.join(ORACLE_DIR, "_ref", "libref_assign.so")
REF_ASSIGN_OFAST_SO = os.path.join(ORACLE_DIR, "_ref", "libref_assign_ofast.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "children_solved", "children_pushed", "dijkstra_steps", "row_visits", "max_queue", "root_steps")]


def build_oracle(force: bool = False) -> None:
    src = os.path.join(ORACLE_DIR, "kbest_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libkbest_oracle.so"], stdout=subprocess.DEVNULL)


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_kbest.restype = C.c_int
        lib.orc_kbest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_double,
                                  _i32p, _i32p, _dp, C.POINTER(OrcStats)]
        lib.orc_kbest_batch.restype = C.c_int64
        lib.orc_kbest_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_double,
                                        _i32p, _i32p, _dp, _i32p, C.c_void_p]
        lib.orc_assign2d.restype = C.c_int
        lib.orc_assign2d.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _i32p, _i32p, _dp]
        lib.orc_assign2d_ex.restype = C.c_int
        lib.orc_assign2d_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _i32p, _i32p, _dp, _dp, _dp]
        lib.orc_condition_costs.restype = C.c_int
        lib.orc_condition_costs.argtypes = [_dp, C.c_int, C.c_int, _dp, _i32p]
        lib.orc_to_probs.restype = None
        lib.orc_to_probs.argtypes = [_dp, C.c_int]
        lib.orc_assignment_prob.restype = C.c_int
        lib.orc_assignment_prob.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dp]
        lib.orc_weights_from_solutions.restype = None
        lib.orc_weights_from_solutions.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _dp, C.c_int, _dp]
        lib.orc_brute_force_prob.restype = C.c_int
        lib.orc_brute_force_prob.argtypes = [_dp, C.c_int, C.c_int, _dp, C.POINTER(C.c_int)]
        lib.orc_permanent.restype = C.c_double
        lib.orc_permanent.argtypes = [_dp, C.c_int, C.c_int]
        lib.orc_quadric_costs.restype = None
        lib.orc_quadric_costs.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp]
        lib.orc_bb_costs.restype = None
        lib.orc_bb_costs.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_double, _dp]
        lib.orc_asgn_bb.restype = None
        lib.orc_asgn_bb.argtypes = [_dp, C.c_int, _dp, C.c_int, C.c_double, _i32p]
        _oracle = lib
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO)


_ref = {}


def ref(ofast: bool = False):
    path = REF_OFAST_SO if ofast else REF_SO
    if path not in _ref:
        lib = C.CDLL(path)
        lib.ref_kbest2d.restype = C.c_int64
        lib.ref_kbest2d.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, _dp, _i64p, _i64p, _dp]
        lib.ref_kbest2d_cutoff.restype = C.c_int64
        lib.ref_kbest2d_cutoff.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, _dp, _i64p, _i64p, _dp,
                                           C.c_double]
        lib.ref_assign2d.restype = C.c_int
        lib.ref_assign2d.argtypes = [C.c_int64, C.c_int64, C.c_int, _dp, _i64p, _i64p, _dp]
        if hasattr(lib, "ref_assign2d_ex"):
            lib.ref_assign2d_ex.restype = C.c_int
            lib.ref_assign2d_ex.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, _dp, _i64p, _i64p, _dp, _dp, _dp]
        lib.ref_kbest2d_batch.restype = C.c_int64
        lib.ref_kbest2d_batch.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _dp,
                                          _i64p, _i64p, _dp, _i64p]
        _ref[path] = lib
    return _ref[path]


def have_ref_assign() -> bool:
    return os.path.exists(REF_ASSIGN_SO)


def ref_assign(ofast: bool = False):
    """The reference's own weights functions (verbatim slices of assignment.cpp, oracle/ref_assign_shim.cpp)."""
    path = REF_ASSIGN_OFAST_SO if ofast else REF_ASSIGN_SO
    if path not in _ref:
        lib = C.CDLL(path)
        lib.ref_assignment_prob.restype = C.c_int
        lib.ref_assignment_prob.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dp]
        lib.ref_brute_force_prob.restype = C.c_int
        lib.ref_brute_force_prob.argtypes = [_dp, C.c_int, C.c_int, _dp]
        lib.ref_condition_costs.restype = C.c_int
        lib.ref_condition_costs.argtypes = [_dp, C.c_int, C.c_int, _dp, _i64p]
        lib.ref_to_probs.restype = None
        lib.ref_to_probs.argtypes = [_dp, C.c_int]
        lib.ref_minc_constant.restype = C.c_double
        lib.ref_minc_constant.argtypes = [C.c_int, C.c_int]
        _ref[path] = lib
    return _ref[path]


def ref_condition_costs(cost, nRows, nCols):
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    out = np.empty(nRows * nCols)
    idx = np.empty(nRows, np.int64)
    good = ref_assign().ref_condition_costs(cost, nRows, nCols, out, idx)
    return out[: good * nCols].copy(), idx[:good].copy()


def ref_assignment_prob(cost, nL, nM, k, ofast=False):
    """Reference assignmentProb: [nM][nL+1] (nM == 1: 1 x len(cost), assignment.cpp:557)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    probs = np.zeros(max(nM * (nL + 1), cost.size) + 8)
    w = ref_assign(ofast).ref_assignment_prob(cost, nL, nM, k, probs)
    return probs[: nM * w].reshape(nM, w).copy()


def ref_brute_force_prob(cost, nL, nM):
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    probs = np.zeros(max(nM * (nL + 1), cost.size) + 8)
    w = ref_assign().ref_brute_force_prob(cost, nL, nM, probs)
    return probs[: nM * w].reshape(nM, w).copy()


# ---------------------------------------------------------------- wrappers

def orc_kbest(cost, N, M, k, maximize=False, cutoff=None, want_stats=False):
    """Oracle k-best of one N x M column-major problem.
    Returns (nf, row4col[k,M], col4row[k,N], gain[k]) (+ stats)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    c4r = np.full((k, N), -7, np.int32)
    r4c = np.full((k, M), -7, np.int32)
    g = np.full(k, np.nan)
    st = OrcStats()
    nf = oracle().orc_kbest(k, N, M, int(maximize), cost, int(cutoff is not None),
                            float(cutoff if cutoff is not None else 0.0), c4r, r4c, g, C.byref(st))
    if want_stats:
        return nf, r4c, c4r, g, st
    return nf, r4c, c4r, g


def orc_kbest_batch(costs, N, M, k, maximize=False, cutoff=None):
    costs = np.ascontiguousarray(costs, dtype=np.float64)
    B = costs.shape[0]
    c4r = np.full((B, k, N), -7, np.int32)
    r4c = np.full((B, k, M), -7, np.int32)
    g = np.full((B, k), np.nan)
    nf = np.zeros(B, np.int32)
    pushed = np.zeros(B, np.int64)
    oracle().orc_kbest_batch(B, k, N, M, int(maximize), costs.reshape(-1), int(cutoff is not None),
                             float(cutoff if cutoff is not None else 0.0), c4r.reshape(-1), r4c.reshape(-1),
                             g.reshape(-1), nf, pushed.ctypes.data)
    return nf, r4c, c4r, g, pushed


def ref_kbest(cost, N, M, k, maximize=False, cutoff=None, ofast=False):
    """The compiled reference itself (only where oracle/_ref exists)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    c4r = np.full((k, N), -7, np.int64)
    r4c = np.full((k, M), -7, np.int64)
    g = np.full(k, np.nan)
    lib = ref(ofast)
    if cutoff is None:
        nf = lib.ref_kbest2d(k, N, M, int(maximize), cost, c4r.reshape(-1), r4c.reshape(-1), g)
    else:
        nf = lib.ref_kbest2d_cutoff(k, N, M, int(maximize), cost, c4r.reshape(-1), r4c.reshape(-1), g,
                                    float(cutoff))
    return int(nf), r4c, c4r, g


def orc_assign2d_ex(cost, N, M, maximize=False, shift=True, gain_cols=0):
    """Oracle assign2D / shortestPathCPP with duals: (ok, row4col[M], col4row[N], gain, u[M], v[N])."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    c4r = np.full(N, -1, np.int32); r4c = np.full(M, -1, np.int32)
    g = np.zeros(1); u = np.zeros(M); v = np.zeros(N)
    ok = oracle().orc_assign2d_ex(N, M, int(maximize), int(shift), gain_cols, cost, c4r, r4c, g, u, v)
    return ok, r4c, c4r, float(g[0]), u, v


def ref_assign2d_ex(cost, N, M, maximize=False, shift=True, gain_cols=0):
    """The compiled reference's assign2D (shift) / shortestPathCPP (no shift) with duals."""
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    c4r = np.full(N, -1, np.int64); r4c = np.full(M, -1, np.int64)
    g = np.zeros(1); u = np.zeros(M); v = np.zeros(N)
    ok = ref().ref_assign2d_ex(N, M, int(maximize), int(shift), gain_cols, cost, c4r, r4c, g, u, v)
    return ok, r4c, c4r, float(g[0]), u, v


def canon_col4row(c4r, M):
    """rows that landed on padded columns (>= M) compare as -1."""
    c = np.array(c4r, dtype=np.int64, copy=True)
    c[c >= M] = -1
    return c


def condition_costs(cost, nRows, nCols):
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    out = np.empty(nRows * nCols)
    idx = np.empty(nRows, np.int32)
    good = oracle().orc_condition_costs(cost, nRows, nCols, out, idx)
    return out[: good * nCols].copy(), idx[:good].copy()


def assignment_prob(cost, nL, nM, k):
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    probs = np.zeros((nM, nL + 1))
    nf = oracle().orc_assignment_prob(cost, nL, nM, k, probs.reshape(-1))
    return probs, nf


def brute_force_prob(cost, nL, nM):
    cost = np.ascontiguousarray(cost, dtype=np.float64).reshape(-1)
    probs = np.zeros((nM, nL + 1))
    uk = C.c_int(0)
    nf = oracle().orc_brute_force_prob(cost, nL, nM, probs.reshape(-1), C.byref(uk))
    return probs, nf, uk.value


def permanent(A):
    """A: (m, n) array.  Exact permanent (rectangular: nwPerm.cpp:217-231 convention)."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    return oracle().orc_permanent(np.ascontiguousarray(A.T).reshape(-1), m, n)


def quadric_costs(m1, cov1, m2, cov2, gate):
    """computeQuadricCostMatrix: m1 (nL,3), cov1 (nL,3,3), m2 (nM,3), cov2 (nM,3,3) -> (nL+nM)*nM column-major."""
    m1 = np.ascontiguousarray(m1, dtype=np.float64); cov1 = np.ascontiguousarray(cov1, dtype=np.float64)
    m2 = np.ascontiguousarray(m2, dtype=np.float64); cov2 = np.ascontiguousarray(cov2, dtype=np.float64)
    nL, nM = len(m1), len(m2)
    out = np.empty((nL + nM) * nM)
    oracle().orc_quadric_costs(m1.reshape(-1), cov1.reshape(-1), nL, m2.reshape(-1), cov2.reshape(-1), nM, float(gate), out)
    return out


def bb_costs(bbL, bbR, gate):
    bbL = np.ascontiguousarray(bbL, dtype=np.float64).reshape(-1, 5); bbR = np.ascontiguousarray(bbR, dtype=np.float64).reshape(-1, 5)
    out = np.empty((len(bbR) + len(bbL)) * len(bbL))
    oracle().orc_bb_costs(bbL.reshape(-1), len(bbL), bbR.reshape(-1), len(bbR), float(gate), out)
    return out


def asgn_bb(bbL, bbR, gate):
    bbL = np.ascontiguousarray(bbL, dtype=np.float64).reshape(-1, 5); bbR = np.ascontiguousarray(bbR, dtype=np.float64).reshape(-1, 5)
    asg = np.full(len(bbL), -1, np.int32)
    oracle().orc_asgn_bb(bbL.reshape(-1), len(bbL), bbR.reshape(-1), len(bbR), float(gate), asg)
    return asg


def canonical_kbest(cost, N, M, k, maximize=False, cutoff=None, cap=4096, run_cap=None):
    """The k best in the engine's ONE order of exact ties (include/kbest_c.h, "Order of exact ties"), from the checker:
    solutions ordered by (gain, row4col lexicographic); when the k-th and the (k+1)-th best gains are equal, the
    lexicographically first assignments of that gain level.  Returns (nf, row4col[nf, M], gain[nf], boundary, resolved):
    boundary = such a tie exists, resolved = its gain level ends within `cap` solutions beyond k (what a synchronous
    entry completes: KBEST_TIE_CAP = 4 096) and -- run_cap, association
    entries only: 4 096 -- has at most run_cap members in all (the longest run of equal gains the device orders, TIE_RUN_CAP).
    The level itself is enumerated completely here, whatever its size."""
    big = k + cap
    while True:
        nf, r4c, c4r, g = orc_kbest(cost, N, M, big, maximize=maximize, cutoff=cutoff)
        if nf < big or nf <= k or g[nf - 1] != g[k - 1]:
            break
        big *= 2
    r4c, g = np.asarray(r4c[:nf]), np.asarray(g[:nf])
    keys = [r4c[:, c] for c in range(M - 1, -1, -1)] + [(-g if maximize else g)]
    order = np.lexsort(keys)  # last key first: gain, then the columns from the first to the last
    boundary = bool(nf > k and g[order[k]] == g[order[k - 1]])
    resolved = boundary and (nf < k + cap or g[order[k + cap - 1]] != g[order[k - 1]]) and (run_cap is None or int((g == g[order[k - 1]]).sum()) <= run_cap)
    n = min(nf, k)
    return n, r4c[order[:n]], g[order[:n]], boundary, resolved


def weights_from_solutions(row4col, gain, nL, nM, gate=True):
    """assignmentProb's accumulation (assignment.cpp:616-648) over a GIVEN ascending list of solutions."""
    r = np.ascontiguousarray(row4col, dtype=np.int32).reshape(-1)
    g = np.ascontiguousarray(gain, dtype=np.float64)
    probs = np.zeros((nM, nL + 1))
    oracle().orc_weights_from_solutions(len(g), nL, nM, r, g, int(bool(gate)), probs.reshape(-1))
    return probs


def canonical_assignment_prob(cond, nL, nM, k, cap=4096):
    """assignmentProb on a conditioned block with the engine's one order of exact ties: the canonical k best within the
    cutoff 42 (canonical_kbest), then the reference's accumulation.  Returns (probs, nf, boundary, resolved)."""
    n, r4c, g, boundary, resolved = canonical_kbest(cond, nL + nM, nM, k, cutoff=42.0, cap=cap, run_cap=4096)
    return weights_from_solutions(r4c, g, nL, nM), n, boundary, resolved
