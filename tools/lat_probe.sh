#!/bin/bash
# tools/lat_probe.sh [tag] -- on the GPU box: per-call latency of the per-frame entry points + kernel trace of the same run
tag=${1:-lat}
out=$(pwd)/gpurun_out/lat_$tag
mkdir -p $out
repo=$(pwd)
python3 tests/dev/bench_latency.py > $out/latency.txt 2>&1
cat $out/latency.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $repo/tests/dev/bench_latency.py > $out/trace.log 2>&1
cd $repo
ls $out/trace
head -20 $out/trace/*kernel_stats.csv
