"""Randomised soak: GPU enumeration against the oracle on many random shapes, flags and cost structures (tests/soak_lib.py).
python tests/dev/soak.py [seconds] [seed].  The -m gpu suite runs a bounded fixed-seed slice of the same generator."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import probabilisticsemslam_amd as pk
import soak_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = pk.KBestEngine(0)
# SOAK_REFERENCE_ORDER=1: KBEST_FLAG_REFERENCE_ORDER, everything slot for slot, ties and padded columns included; =2: the DEFAULT of the
# synchronous entry, the fast kernels + the tied problems again on the reference-order kernel: gains and row4col slot for slot on every
# problem; unset: KBEST_FLAG_CANONICAL_TIES, the engine's own rule on ties (every kernel and the completion of tied levels)
ref_order = {"1": True, "2": 2}.get(os.environ.get("SOAK_REFERENCE_ORDER", ""), False)
ncase, nprob, bad = soak_lib.run(eng, seed, seconds=budget, big_frac=float(os.environ.get("SOAK_BIG", "0.12")),
                                 big_max=int(os.environ.get("SOAK_BIGMAX", "200")), reference_order=ref_order)
if bad:
    print("MISMATCH", bad); sys.exit(1)
print(f"soak ok: {ncase} cases, {nprob} problems (seed {seed}); {eng.relay_launches()} launches of the 64-row kernel were relays")
