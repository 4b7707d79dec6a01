"""Randomised soak of the association path (cost block in -> probabilities out) against the reference's own
conditionCosts + assignmentProb as restated by the oracle.  python tests/dev/soak_assoc.py [seconds] [seed].
SOAK_ASSOC_REFERENCE=1 / 2: kbest_set_reference_order(ctx, 1 / 2) and INTEGER costs on the gated entries (masses of exact ties, many of them
across slot k): the probabilities must then be the reference's own -- the heap's choice among a tied level -- on every frame.
Development aid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probabilisticsemslam_amd as pk
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = pk.KBestEngine(0)
ref_mode = int(os.environ.get("SOAK_ASSOC_REFERENCE", "0"))
if ref_mode:
    eng.set_reference_order(ref_mode)
t0 = time.time(); ncase = nfr = 0; worst = 0.0; rerun = 0
use_ref = False  # the oracle (pinned to the compiled reference by tests/golden/weights_golden.npz) has one output shape for every nM
while time.time() - t0 < budget:
    F = int(rng.choice([1, 3, 17, 64]))
    frames, nLs, nMs = [], [], []
    for _ in range(F):
        nL = int(rng.integers(1, 70)); nM = int(rng.integers(1, min(13, nL + 1) + 1))
        nM = min(nM, nL) if rng.random() < 0.9 else nM
        nR = nL + nM
        C = np.full(nR * nM, np.inf)
        p = rng.random() * 0.3 + 2.0 / nL
        for c in range(nM):
            for r in range(nL):
                if rng.random() < p or r == c: C[c * nR + r] = float(rng.integers(0, 9)) if ref_mode else 12.0 * rng.random() * rng.random()
                else: C[c * nR + r] = 60.0 + 400.0 * rng.random()
            C[c * nR + nL + c] = 10.0
        frames.append(C); nLs.append(nL); nMs.append(nM)
    k = int(rng.choice([1, 20, 200, 200, 1000]))
    probs, nf = eng.weights(frames, nLs, nMs, k, condition=True)
    if ref_mode == 2:
        rerun += int((eng.last_tie_flags() & pk.engine.KBEST_TIE_REFERENCE).astype(bool).sum())
    for f in range(F):
        nL, nM = nLs[f], nMs[f]
        cond, idx = (ol.ref_condition_costs if use_ref else ol.condition_costs)(frames[f], nL + nM, nM)
        nLc = len(idx) - nM
        if nLc < 0: continue  # undefined in the reference (size_t underflow)
        if use_ref: pc = ol.ref_assignment_prob(cond, nLc, nM, k)[0]
        else: pc = ol.assignment_prob(cond, nLc, nM, k)[0]
        want = np.zeros((nM, nL + 1))
        for j in range(nLc): want[:, idx[j]] = pc[:, j]
        want[:, nL] = pc[:, nLc]
        err = float(np.abs(probs[f] - want).max())
        worst = max(worst, err)
        if not err < 1e-10:
            print("MISMATCH", dict(nL=nL, nM=nM, k=k, F=F, f=f, err=err, seed=seed)); sys.exit(1)
    ncase += 1; nfr += F
mode = {0: "", 1: "; kbest_set_reference_order(1), integer costs", 2: f"; kbest_set_reference_order(2), integer costs: {rerun} frames run again on the reference-order kernel"}[ref_mode]
print(f"assoc soak ok: {ncase} calls, {nfr} frames in {time.time() - t0:.0f} s (seed {seed}), worst abs err {worst:.2e}, reference = {'compiled' if use_ref else 'oracle'}{mode}")
