#!/bin/bash
# tools/collect_profiles.sh <run-tag> -- copy the summaries of gpurun_out/prof_<run-tag>_<cfg>/ into profiles/r02_<cfg>_*
tag=${1:-r02f}
for c in c4 c2 c3 c5 w128; do
  d=gpurun_out/prof_${tag}_$c
  [ -d $d ] || continue
  cp $d/summary.json profiles/r02_${c}_summary.json
  cp $d/summary.txt profiles/r02_${c}_rocprof_summary.txt
  cp $d/trace/trace_kernel_stats.csv profiles/r02_${c}_kernel_stats.csv
done
