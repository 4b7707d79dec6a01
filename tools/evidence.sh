#!/bin/bash
# tools/evidence.sh <round> [soak seconds] -- on the GPU box: everything profiles/<round>_* is made from, in one call: the rocprofv3
# kernel trace + PMC passes of every config (tools/prof.sh), the bench line, the one-frame-per-call crossover, the soaks.
# Afterwards, in the build container:  tools/collect_profiles.sh <round> <round>; cp gpurun_out/<round>_bench_line.json
# gpurun_out/<round>_crossover.json gpurun_out/<round>_soak/soak.log profiles/ ...  (see profiles/README.md).
rnd=${1:-r05}; secs=${2:-60}
for c in c4 c2 c3 c5 w128; do bash tools/prof.sh ${rnd}_$c $c > /dev/null 2>&1; tail -3 gpurun_out/prof_${rnd}_$c/summary.txt; done
python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${rnd}_bench_line.json
python3 tests/dev/crossover.py gpurun_out/${rnd}_crossover.json > /dev/null 2>&1
bash tools/soak.sh ${rnd}_soak $secs 900 > /dev/null 2>&1
tail -6 gpurun_out/${rnd}_soak/soak.log
