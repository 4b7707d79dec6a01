"""Print the last rows of a rocprofv3 kernel + memory-copy trace as one timeline (ms).  usage: timeline.py <dir> [rows]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:48]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
if rows:
    t0 = rows[-n][0] if len(rows) >= n else rows[0][0]
    for s, e, name in rows[-n:]:
        print("%10.3f %10.3f  %8.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, name))
