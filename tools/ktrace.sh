#!/bin/bash
# tools/ktrace.sh <script.py> [args] -- on the GPU box: rocprofv3 kernel trace of one python script, per-kernel table.
repo=$(pwd); out=$repo/gpurun_out/ktrace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $repo/"$@" > $out/run.log 2>&1
tail -6 $out/run.log | grep -v simple_timer
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])): print(r["Name"][:60].ljust(60),r["Calls"].rjust(6),r["AverageNs"].rjust(12),r["MaxNs"].rjust(12),r["Percentage"])
PY
