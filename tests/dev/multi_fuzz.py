"""Dev aid (GPU box): the one-process multi-device entry on LOGICAL devices (the same GPU several times) under random batches -- batch
mode and subtree mode (gains first / whole lists, S shards over G devices), uniform and ragged shapes, integer (exact ties) and
continuous costs, maximise, cutoff, with and without KBEST_FLAG_CANONICAL_TIES -- against the checker: counts, gains (bits), valid
assignments; row4col slot for slot where the checker's k + 1 best gains are all different (or reference ties were asked for); every
device's global table equal.  usage: python tests/dev/multi_fuzz.py [seconds] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
multis = {G: pk.KBestMulti([0] * G) for G in (1, 2, 3, 8)}
t0 = time.time()
ncall = nsub = 0
while time.time() - t0 < budget:
    G = int(rng.choice([1, 2, 3, 8]))
    m = multis[G]
    subtree = bool(rng.random() < 0.4)
    B = int(rng.choice([1, 2, 5, 17, 40])) if subtree else int(rng.choice([1, 3, 8, 33, 70]))
    N = int(rng.integers(2, 41)) if rng.random() < 0.85 else int(rng.integers(65, 140))
    M = int(rng.integers(1, N + 1)) if rng.random() < 0.5 else N
    k = int(rng.choice([1, 3, 20, 90]))
    integer = bool(rng.random() < 0.5)
    costs = rng.integers(0, 6, (B, N * M)).astype(np.float64) if integer else rng.random((B, N * M))
    kw = dict(maximize=bool(rng.random() < 0.2), cutoff=(float(rng.random() * 3) if rng.random() < 0.25 else None))
    ragged = (not subtree) and rng.random() < 0.3
    nRow = nCol = None
    if ragged:
        nRow = rng.integers(1, N + 1, B).astype(np.int32)
        nCol = np.array([int(rng.integers(1, min(int(r), M) + 1)) for r in nRow], np.int32)
        nRow[0], nCol[0] = N, M
    ref_ties = (not subtree) and rng.random() < 0.7  # (batch mode: the reference's answer on ties is the default; else the engine's rule)
    n_shard = int(rng.choice([0, 2, 5, 16])) if subtree else 0
    nf, r4c, c4r, g = m.kbest(costs, N, M, k, nRow=nRow, nCol=nCol, subtree=subtree, n_shard=n_shard, canonical_ties=not ref_ties, **kw)
    desc = (seed, ncall, G, "subtree" if subtree else "batch", n_shard, B, N, M, k, integer, ragged, ref_ties, kw)
    assert m.tables_agree(), ("devices hold different global tables", desc)
    for b in range(B):
        n_, m_ = (int(nRow[b]), int(nCol[b])) if ragged else (N, M)
        blk = costs[b, : n_ * m_]
        wn, wr, wc, wg = ol.orc_kbest(blk, n_, m_, min(k + 1, 10 ** 9), **kw)
        tie_free = wn < 2 or not (wg[1:wn] == wg[: wn - 1]).any()
        wn = min(wn, k)
        assert nf[b] == wn, ("nf", desc, b, int(nf[b]), wn)
        assert (g[b, :wn].view(np.int64) == wg[:wn].view(np.int64)).all(), ("gain", desc, b)
        for s in range(wn):
            rows = r4c[b, s, :m_].astype(np.int64)
            assert len(set(rows.tolist())) == m_ and rows.min() >= 0 and rows.max() < n_, ("assignment", desc, b, s)
        if tie_free or ref_ties:
            assert (r4c[b, :wn, :m_] == wr[:wn]).all(), ("row4col", desc, b)
    ncall += 1
    nsub += int(subtree)
for m in multis.values():
    m.close()
print(f"multi fuzz ok: {ncall} calls ({nsub} in subtree mode) in {budget:.0f} s (seed {seed})")
