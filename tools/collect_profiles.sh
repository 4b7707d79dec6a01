#!/bin/bash
# tools/collect_profiles.sh <run-tag> [round] -- copy the summaries of gpurun_out/prof_<run-tag>_<cfg>/ into profiles/<round>_<cfg>_*
tag=${1:-r03f}
rnd=${2:-r03}
for c in c4 c2 c3 c5 w128; do
  d=gpurun_out/prof_${tag}_$c
  [ -d $d ] || continue
  cp $d/summary.json profiles/${rnd}_${c}_summary.json
  cp $d/summary.txt profiles/${rnd}_${c}_rocprof_summary.txt
  cp $d/trace/trace_kernel_stats.csv profiles/${rnd}_${c}_kernel_stats.csv
done
