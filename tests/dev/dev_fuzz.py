"""Dev aid (GPU box): the ASYNCHRONOUS entry (kbest_batch_f64_dev) + its second call (kbest_resolve_ties_dev) under random batches --
uniform and ragged shapes, int8 tables, integer (exact ties) and continuous costs, maximise, cutoff.  By default (the reference's answer on
ties) the device tables must end up equal to the checker's, slot for slot; with KBEST_FLAG_CANONICAL_TIES (the engine's own rule) tables and
flags equal to the synchronous host entry's under the same flag.
usage: python tests/dev/dev_fuzz.py [seconds] [seed]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import probabilisticsemslam_amd as pk
import oracle_lib as ol

E = pk.engine
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = pk.KBestEngine(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
t0 = time.time()
ncall = nref = 0
while time.time() - t0 < budget:
    B = int(rng.choice([1, 2, 9, 40, 300]))
    maxN = int(rng.integers(1, 41)) if rng.random() < 0.9 else int(rng.integers(65, 120))
    maxM = int(rng.integers(1, maxN + 1)) if rng.random() < 0.5 else maxN
    k = int(rng.choice([1, 2, 7, 30, 120]))
    i8 = bool(rng.random() < 0.4)
    ragged = bool(rng.random() < 0.4)
    nRow = rng.integers(1, maxN + 1, B).astype(np.int32) if ragged else np.full(B, maxN, np.int32)
    nCol = np.array([int(rng.integers(1, min(int(r), maxM) + 1)) for r in nRow], np.int32) if ragged else np.full(B, maxM, np.int32)
    if ragged:
        nRow[0], nCol[0] = maxN, maxM
    integer = bool(rng.random() < 0.6)
    blocks = [(rng.integers(0, 5, int(r) * int(c)).astype(np.float64) if integer else rng.random(int(r) * int(c))) for r, c in zip(nRow, nCol)]
    off = np.zeros(B, np.int64)
    off[1:] = np.cumsum([len(x) for x in blocks[:-1]])
    flat = np.concatenate(blocks)
    kw = dict(maximize=bool(rng.random() < 0.2), cutoff=(float(rng.random() * 3) if rng.random() < 0.25 else None))
    ref = bool(rng.random() < 0.4)
    tdt = torch.int8 if i8 else torch.int32
    d_cost = torch.from_numpy(flat).to(dev)
    d_r = torch.full((B, k, maxM), -5, dtype=tdt, device=dev)
    d_c = torch.full((B, k, maxN), -5, dtype=tdt, device=dev)
    d_g = torch.zeros((B, k), dtype=torch.float64, device=dev)
    d_n = torch.full((B,), -5, dtype=torch.int32, device=dev)
    d_f = torch.zeros(B, dtype=torch.int32, device=dev)
    shp = dict(d_nRow=torch.from_numpy(nRow).to(dev), d_nCol=torch.from_numpy(nCol).to(dev), d_costOff=torch.from_numpy(off).to(dev)) if ragged else {}
    torch.cuda.synchronize()
    eng.reserve(B, maxN, k)
    if ref:
        eng.lib.kbest_reserve_exact(eng.ctx, B, maxN, maxM, k)
    eng.kbest_dev(d_cost, B, maxN, maxM, k, d_r, d_c, d_g, d_n, stream=st, d_tie_flags=d_f, tables_i8=i8, **kw, **shp)
    eng.resolve_ties_dev(d_cost, B, maxN, maxM, k, d_r, d_c, d_g, d_f, stream=st, tables_i8=i8, canonical_ties=not ref, **kw, **shp)
    torch.cuda.synchronize()
    nf, r4c, c4r, g, fl = d_n.cpu().numpy(), d_r.cpu().numpy().astype(np.int32), d_c.cpu().numpy().astype(np.int32), d_g.cpu().numpy(), d_f.cpu().numpy()
    desc = (seed, ncall, B, maxN, maxM, k, i8, ragged, integer, ref, kw)
    if ref:
        for b in range(B):
            n_, m_ = int(nRow[b]), int(nCol[b])
            wn, wr, wc, wg = ol.orc_kbest(blocks[b], n_, m_, k, **kw)
            assert nf[b] == wn, ("nf", desc, b)
            assert (g[b, :wn].view(np.int64) == wg[:wn].view(np.int64)).all() and (r4c[b, :wn, :m_] == wr[:wn]).all(), ("tables", desc, b, hex(int(fl[b])))
            assert not (fl[b] & (E.KBEST_TIE_BOUNDARY | E.KBEST_TIE_UNRESOLVED)), ("flags", desc, b, hex(int(fl[b])))
        nref += 1
    else:
        if ragged:
            want = eng.kbest(flat, maxN, maxM, k, nRow=nRow, nCol=nCol, costOff=off, tie_flags=True, tables_i8=i8, canonical_ties=True, **kw)
        else:
            want = eng.kbest(flat.reshape(B, -1), maxN, maxM, k, tie_flags=True, tables_i8=i8, canonical_ties=True, **kw)
        assert (nf == want[0]).all(), ("nf", desc)
        # (the asynchronous entry may run WITHOUT the extra solution where k sits at a kernel's limit: KBEST_TIE_UNCHECKED there)
        same = (fl & ~E.KBEST_TIE_UNCHECKED) == want[4]
        assert same.all(), ("flags", desc, [hex(int(x)) for x in fl[~same][:4]], [hex(int(x)) for x in want[4][~same][:4]])
        for b in range(B):
            if fl[b] & (E.KBEST_TIE_UNRESOLVED | E.KBEST_TIE_UNCHECKED):
                continue
            n_, m_ = int(nRow[b]), int(nCol[b])
            n = int(nf[b])
            assert (g[b, :n].view(np.int64) == want[3][b, :n].view(np.int64)).all() and (r4c[b, :n, :m_] == want[1][b, :n, :m_]).all(), ("tables", desc, b, hex(int(fl[b])))
    ncall += 1
print(f"dev fuzz ok: {ncall} calls ({nref} with the reference's ties -- the default --, the others with the engine's rule) in {budget:.0f} s (seed {seed})")
