#!/bin/bash
# tools/r04_final2.sh -- on the GPU box: evidence of the round's LAST build for what changed after tools/r04_final.sh ran (the
# association path: exhaustive kernel, bounded walk): tests, soak, the C5 profile, the bench line, the crossover table
out=$(pwd)/gpurun_out/r04_final2
mkdir -p $out
timeout 1200 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $out/tests.txt
{
echo "# last build of the round: the association path (kbest_tiny.hip, kbest_bnb.hip) against the enumeration kernels, and the rest once more"
echo "exhaustive kernel + bounded walk against the enumeration kernels (SOAK_BNB=1: up to 16 measurements, 64 rows), 240 s, seed 71: $(SOAK_BNB=1 timeout 500 python3 tests/dev/soak_tiny.py 240 71 2>&1 | tail -1)"
echo "the same, 120 s, seed 72: $(SOAK_BNB=1 timeout 400 python3 tests/dev/soak_tiny.py 120 72 2>&1 | tail -1)"
echo "association path against the checker, 180 s, seed 73: $(timeout 400 python3 tests/dev/soak_assoc.py 180 73 2>&1 | tail -1)"
echo "default routing, 120 s, seed 74: $(timeout 300 python3 tests/dev/soak.py 120 74 2>&1 | tail -1)"
} > $out/soak3.log 2>&1
cat $out/soak3.log
bash tools/prof.sh r04f_c5 c5 > $out/prof_c5.log 2>&1
tail -3 $out/prof_c5.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err
tail -c 300 $out/bench_line.json
timeout 600 python3 tests/dev/crossover.py $out/crossover.json > $out/crossover.log 2>&1
tail -5 $out/crossover.log
