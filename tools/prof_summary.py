"""Summarise the rocprofv3 CSVs written by tools/prof.sh (per-kernel durations and PMC sums)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel durations (us) from --kernel-trace")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} n={len(v):4d} avg={sum(v)/len(v):12.1f} min={min(v):12.1f} max={max(v):12.1f} total={sum(v):12.1f}")
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(f"{d}/**/*counter_collection.csv"):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"== {d}: per-dispatch mean of each counter")
    for k, cs in acc.items():
        if "kbest" not in k:
            continue
        for c, v in sorted(cs.items()):
            print(f"{k:40s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
