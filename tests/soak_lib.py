"""Randomised comparison of the HIP enumeration with the oracle: random shapes, batch sizes, maximise / cutoff and seven
cost structures (uniform, small integers = exact ties, +inf patterns, planted cheap swaps, near-ties at 1e-9, scaled,
negative).  Used by the bounded fixed-seed soak of the -m gpu suite and by tests/dev/soak.py (minutes).  Test
infrastructure: the engine under test goes through the C ABI, the oracle is only the checker."""
import time

import numpy as np

import oracle_lib as ol

KINDS = ["uniform", "ints", "inf", "blocks", "near", "scaled", "neg"]


def draw_case(rng, max_rows=64, big_frac=0.0, big_max=200, batches=(1, 2, 5, 9)):
    big = rng.random() < big_frac
    N = int(rng.integers(65, big_max)) if big else int(rng.integers(1, max_rows + 1))
    M = int(rng.integers(1, N + 1)) if rng.random() < 0.6 else N
    k = int(rng.choice([1, 2, 3, 7, 50, 200, 300])) if not big else int(rng.choice([3, 20, 60]))
    B = int(rng.choice(batches))
    maximize = bool(rng.random() < 0.2)
    cutoff = float(rng.random() * 2) if rng.random() < 0.25 else None
    kind = KINDS[int(rng.integers(len(KINDS)))]
    C = rng.random((B, N * M))
    if kind == "ints":
        C = rng.integers(0, 4, (B, N * M)).astype(np.float64)
    elif kind == "inf":
        C[rng.random((B, N * M)) < rng.random() * 0.8] = np.inf
    elif kind == "blocks":
        C = C * 5 + 10
        for b in range(B):
            Cm = C[b].reshape(M, N)
            for c in range(M):
                Cm[c, (c * 7) % N] = rng.random() * 0.05
                Cm[c, (c * 7 + 1) % N] = rng.random() * 0.05
    elif kind == "near":
        C = 1.0 + C * 1e-9
    elif kind == "scaled":
        C = C * 1e6
    elif kind == "neg":
        C = C - 0.5
    if maximize and kind == "inf":
        C = np.where(np.isinf(C), -np.inf, C)
    return dict(N=N, M=M, k=k, B=B, maximize=maximize, cutoff=cutoff, kind=kind, C=C)


def check_case(eng, case, reference_order=False):
    """None if the engine's result equals the oracle's (nf, gains bit for bit, assignments; equal gains as multisets),
    else a description of the first mismatch.  reference_order: KBEST_FLAG_REFERENCE_ORDER -- then EVERYTHING must be the
    checker's, slot for slot: the order of equal gains and col4row on padded columns included."""
    N, M, k, C = case["N"], case["M"], case["k"], case["C"]
    # every third case takes its tables as int8 (KBEST_FLAG_TABLES_I8): the same values from every kernel's output phase
    i8 = N <= 127 and (N + M + k + case["B"]) % 3 == 0
    # reference_order == 2: the synchronous entry's DEFAULT -- the fast kernels, the tied problems again on the reference-order kernel:
    # gains and row4col the checker's slot for slot on EVERY problem, col4row up to the names of padded columns (raw on the re-run
    # problems).  reference_order False: KBEST_FLAG_CANONICAL_TIES -- the engine's own rule on ties, every kernel and its completion of
    # tied levels (equal gains compared as multisets below)
    ties_only = reference_order == 2
    nf, r4c, c4r, g = eng.kbest(C, N, M, k, case["maximize"], case["cutoff"], tables_i8=i8, reference_order=reference_order is True or reference_order == 1,
                                canonical_ties=reference_order is False)[:4]
    onf, or4c, oc4r, og, _ = ol.orc_kbest_batch(C, N, M, k, case["maximize"], case["cutoff"])
    for b in range(case["B"]):
        n = int(onf[b])
        ok = nf[b] == n and (g[b, :n].view(np.int64) == og[b, :n].view(np.int64)).all()
        if ties_only:
            named = lambda t: np.where(t >= M, -1, t)  # noqa: E731
            ok = ok and (r4c[b, :n] == or4c[b, :n]).all() and (named(c4r[b, :n].astype(np.int32)) == named(oc4r[b, :n])).all()
            if not ok:
                return dict(b=b, nf=int(nf[b]), onf=n, reference_ties=True, **{key: case[key] for key in ("N", "M", "k", "B", "maximize", "cutoff", "kind")})
            continue
        if reference_order:
            ok = ok and (r4c[b, :n] == or4c[b, :n]).all() and (c4r[b, :n] == oc4r[b, :n]).all()
            if not ok:
                return dict(b=b, nf=int(nf[b]), onf=n, reference_order=True, **{key: case[key] for key in ("N", "M", "k", "B", "maximize", "cutoff", "kind")})
            continue
        if ok and not (r4c[b, :n] == or4c[b, :n]).all():
            # equal gains may come out in another order: assignments as multisets below the last gain
            got = sorted((float(g[b, i]), tuple(r4c[b, i].tolist())) for i in range(n))
            want = sorted((float(og[b, i]), tuple(or4c[b, i].tolist())) for i in range(n))
            last = float(og[b, n - 1]) if n else 0.0
            ok = [x for x in got if x[0] != last] == [x for x in want if x[0] != last] and len({x[1] for x in got}) == n
        if not ok:
            return dict(b=b, nf=int(nf[b]), onf=n, **{key: case[key] for key in ("N", "M", "k", "B", "maximize", "cutoff", "kind")})
    return None


def run(eng, seed, n_cases=None, seconds=None, reference_order=False, **draw_kw):
    """Fixed number of cases (reproducible) or a time budget.  Returns (cases, problems, first mismatch or None)."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    ncase = nprob = 0
    while (n_cases is None or ncase < n_cases) and (seconds is None or time.time() - t0 < seconds):
        case = draw_case(rng, **draw_kw)
        bad = check_case(eng, case, reference_order)
        if bad is not None:
            bad["seed"] = seed
            bad["case_no"] = ncase
            return ncase, nprob, bad
        ncase += 1
        nprob += case["B"]
    return ncase, nprob, None
