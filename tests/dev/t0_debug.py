"""Debug aid: print the a-priori thresholds (KB_T0_DEBUG build) next to the true gap of the first matrices of C4."""
import os, sys
import numpy as np
os.environ["KBEST_LIB"] = "libkbest_amd_dbg.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import probabilisticsemslam_amd as pk
from probabilisticsemslam_amd import workloads as wl
eng = pk.KBestEngine(0)
costs, N, M, k = wl.dense_config("c4", B=512)
nf, r4c, c4r, g = eng.kbest(costs, N, M, k)[:4]
for b in range(4):
    print(f"matrix {b}: true gap {g[b, k-1] - g[b, 0]:.6f}")
