"""Dev loop (GPU box): the bounded-walk kernel (kbest_bnb.hip) against the enumeration kernel (KBEST_NO_BNB) through
kbest_assoc_probs_batch_f64: equality of counts and probabilities on KITTI-like and dense frames, one-frame and batched time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl

bnb = pk.KBestEngine(0)
os.environ["KBEST_NO_BNB"] = "1"
plain = pk.KBestEngine(0)
rng = np.random.default_rng(3)

def dense_frames(F, nL, nM):
    out = []
    for _ in range(F):
        nR = nL + nM
        C = np.full(nR * nM, np.inf)
        for c in range(nM):
            C[c * nR: c * nR + nL] = rng.random(nL) * 30.0
            C[c * nR + nL + c] = 10.0
        out.append(C)
    return out

def timeit(eng, frames, nL, nM, k=200, n=100):
    for i in range(-30, n):
        if i == 0:
            t0 = time.perf_counter()
        eng.weights([frames[i % len(frames)]], [nL], [nM], k, condition=True)
    return 1e6 * (time.perf_counter() - t0) / n

for (nL, nM, kind) in ((20, 10, "kitti"), (20, 10, "dense"), (40, 12, "kitti"), (12, 5, "kitti"), (30, 16, "kitti"), (50, 8, "kitti")):
    fr = wl.kitti_like_frames(48, nL=nL, nM=nM, seed=5 + nL) if kind == "kitti" else dense_frames(16, nL, nM)
    bad = 0
    for k in (200, 7):
        out, nf = bnb.weights(fr, [nL] * len(fr), [nM] * len(fr), k, condition=True)
        ref, nfr = plain.weights(fr, [nL] * len(fr), [nM] * len(fr), k, condition=True)
        bad += sum(int(nf[i] != nfr[i] or not np.array_equal(out[i], ref[i])) for i in range(len(fr)))
    print(f"{kind} {nL}x{nM}: mismatching frames {bad}/{2*len(fr)}; one frame per call: bounded walk {timeit(bnb, fr, nL, nM):.1f} us, enumeration {timeit(plain, fr, nL, nM):.1f} us (python overhead included)", flush=True)
F = 1000
fr = wl.kitti_like_frames(F)
for eng, name in ((bnb, "bounded walk"), (plain, "enumeration")):
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        out, nf = eng.weights(fr, [20] * F, [10] * F, 200, condition=True)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{F} frames per call, host-inclusive (python packing included): {name} {min(ts):.3f} ms")
