"""Dev aid (GPU box): random COMBINATIONS of the host entry's options -- int8 tables, tie flags, tie check / resolve off, the reference's
order, reference ties, push counting, no pruning, ragged shapes, cutoff, maximise -- on small integer / continuous problems.  Every call
must come back without an error, with the checker's counts and gains (bits) and with valid assignments whose serial gain is the gain
reported.  usage: python tests/dev/combo_fuzz.py [seconds] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = pk.KBestEngine(0)
t0 = time.time()
ncall = 0
while time.time() - t0 < budget:
    B = int(rng.choice([1, 2, 7, 33]))
    maxN = int(rng.integers(1, 41))
    maxM = int(rng.integers(1, maxN + 1))
    k = int(rng.choice([1, 2, 5, 40, 130]))
    ragged = rng.random() < 0.4
    nRow = rng.integers(1, maxN + 1, B).astype(np.int32) if ragged else np.full(B, maxN, np.int32)
    nCol = np.array([int(rng.integers(1, min(int(r), maxM) + 1)) for r in nRow], np.int32) if ragged else np.full(B, maxM, np.int32)
    if ragged:
        nRow[0], nCol[0] = maxN, maxM
    integer = rng.random() < 0.6
    blocks = [(rng.integers(0, 5, int(r) * int(c)).astype(np.float64) if integer else rng.random(int(r) * int(c))) for r, c in zip(nRow, nCol)]
    if rng.random() < 0.2:
        blocks[-1][: int(nRow[-1])] = np.inf  # an infeasible problem (its first column is all +inf)
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum([len(x) for x in blocks[:-1]])
    kw = dict(maximize=bool(rng.random() < 0.2), cutoff=(float(rng.random() * 3) if rng.random() < 0.3 else None))
    if kw["maximize"]:
        blocks = [np.where(np.isinf(x), -np.inf, x) for x in blocks]
    opt = dict(tables_i8=bool(rng.random() < 0.3), tie_flags=bool(rng.random() < 0.5), tie_check=bool(rng.random() < 0.85),
               tie_resolve=bool(rng.random() < 0.85), count_pushed=bool(rng.random() < 0.15), prune=bool(rng.random() < 0.85))
    mode = rng.choice(["default", "canonical", "order", "ties"])  # (default = the reference's answer on ties; "ties": the accepted flag)
    if mode == "order":
        opt["reference_order"] = True
    elif mode == "ties":
        opt["reference_ties"] = True
    elif mode == "canonical":
        opt["canonical_ties"] = True
    flat = np.concatenate(blocks)
    if ragged:
        out = eng.kbest(flat, maxN, maxM, k, nRow=nRow, nCol=nCol, costOff=off, **kw, **opt)
    else:
        out = eng.kbest(flat.reshape(B, -1), maxN, maxM, k, **kw, **opt)
    nf, r4c, c4r, g = out[:4]
    for b in range(B):
        n_, m_ = int(nRow[b]), int(nCol[b])
        wn, wr, wc, wg = ol.orc_kbest(blocks[b], n_, m_, k, **kw)
        desc = (seed, ncall, b, n_, m_, k, mode, kw, opt, integer, ragged)
        assert nf[b] == wn, ("nf", desc, int(nf[b]), wn)
        assert (g[b, :wn].view(np.int64) == wg[:wn].view(np.int64)).all(), ("gain", desc)
        Cm = blocks[b].reshape(m_, n_)
        for s in range(wn):
            rows = r4c[b, s, :m_].astype(np.int64)
            assert len(set(rows.tolist())) == m_ and rows.min() >= 0 and rows.max() < n_, ("assignment", desc, s)
            inv = c4r[b, s, :n_].astype(np.int64)
            assert all(inv[rows[c]] == c for c in range(m_)), ("col4row", desc, s)
        if mode != "canonical" and opt["tie_check"] and opt["tie_resolve"] and not opt["count_pushed"] and opt["prune"] or mode == "order":
            assert (r4c[b, :wn, :m_] == wr[:wn]).all(), ("reference order", desc)
    ncall += 1
print(f"combo fuzz ok: {ncall} calls in {budget:.0f} s (seed {seed})")
