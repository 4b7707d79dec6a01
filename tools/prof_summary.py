"""Summarise the rocprofv3 CSVs written by tools/prof.sh: per-kernel durations, PMC means, HBM traffic.
The bench run under the profiler makes 1 untimed no-prune dispatch (push counting), W warm-up dispatches and K timed
ones; only the last K dispatches of the k-best kernel are the measured workload, so means are taken over those."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = sys.argv[3] if len(sys.argv) > 3 else "c4"
BATCH = {"c2": 1024, "c3": 4096, "c4": 1024, "c5": 1000, "w128": 512}.get(cfg)


def rows(pattern):
    for f in sorted(glob.glob(os.path.join(out, pattern), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"][:60]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("== kernel durations (us) from --kernel-trace --stats")
res = {"config": cfg, "batch": BATCH}
# the measured kernel = the k-best kernel of the LAST dispatch in the trace (the timed launches come last; the untimed push-counting
# launch in front of them runs another kernel and can outweigh six short timed launches in total time)
cands = [k for k in dur if "kbest" in k and "fill" not in k and "merge" not in k]
main = max(cands, key=lambda k: max(t for t, _ in dur[k]), default=None)
if main is not None:
    # ... or the kernel that shares the timed launches with it and takes most of their time (the association entry follows the
    # bounded walk by a launch of the enumeration kernel that only looks for frames handed back: microseconds)
    t_from = sorted(t for t, _ in dur[main])[-K:][0]
    # (a kernel that ran ONCE in front of the timed launches -- the untimed push-counting launch -- is not part of them, however long it took)
    timed = lambda k: sum(d for t, d in sorted(dur[k])[-K:] if t >= t_from - 5_000_000) if sum(1 for t, _ in dur[k] if t >= t_from - 5_000_000) >= K else 0.0
    main = max(cands, key=timed)
res["kernel"] = main
for k, v in sorted(dur.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    d = [x for _, x in v]
    line = f"{k:60s} n={len(d):4d} avg={sum(d)/len(d):12.1f} min={min(d):12.1f} max={max(d):12.1f} total={sum(d):12.1f}"
    if k == main:
        last = d[-K:]
        line += f"  | last {K} (timed workload): avg={sum(last)/len(last):.1f}"
        res["kernel_avg_us_timed"] = sum(last) / len(last)
    print(line)
pm = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(f"{d}/**/*counter_collection.csv"):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    if acc:
        print(f"== {d}: mean over the last {K} dispatches of the k-best kernel")
    for k, cs in acc.items():
        if main is None or k[:40] != main[:40]:
            continue
        for c, v in sorted(cs.items()):
            v.sort()
            last = [x for _, x in v][-K:]
            pm[c] = sum(last) / len(last)
            print(f"{k:40s} {c:24s} n={len(last):3d} mean={pm[c]:.6g}")
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a
    # coalesced stream -> double it; WRITE_SIZE reads exactly.  Separate --pmc passes (TCC slots).
    traffic = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
    print(f"== HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 = {traffic:.4g} bytes "
          f"(FETCH_SIZE {pm['FETCH_SIZE']:.6g} KiB, WRITE_SIZE {pm['WRITE_SIZE']:.6g} KiB)")
    res.update({"bytes_per_launch": traffic, "FETCH_SIZE_KiB": pm["FETCH_SIZE"], "WRITE_SIZE_KiB": pm["WRITE_SIZE"],
                "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM]"})
res["pmc"] = pm
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
