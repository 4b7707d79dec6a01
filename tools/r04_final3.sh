#!/bin/bash
# tools/r04_final3.sh -- on the GPU box: evidence of the round's LAST build (bounded walk lowering its bound in place; weights
# accumulated by a wave-uniform walk over the solutions in all three association kernels): the whole GPU suite, soaks of the
# association path, the C5 profile, the bench line, the crossover table
out=$(pwd)/gpurun_out/r04_final3
mkdir -p $out
timeout 1200 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $out/tests.txt
{
echo "# the round's last build (weights epilogues of kbest_bnb / kbest_tiny / kbest_small reworked; commit 01c7cb2 and the one after)"
echo "exhaustive kernel + bounded walk against the enumeration kernels (SOAK_BNB=1: up to 16 measurements, 64 rows), 90 s, seed 141: $(SOAK_BNB=1 timeout 400 python3 tests/dev/soak_tiny.py 90 141 2>&1 | tail -1)"
echo "association path against the checker, 60 s, seed 142: $(timeout 300 python3 tests/dev/soak_assoc.py 60 142 2>&1 | tail -1)"
echo "association path on the enumeration kernels only (KBEST_NO_TINY KBEST_NO_BNB) against the checker, 60 s, seed 143: $(KBEST_NO_TINY=1 KBEST_NO_BNB=1 timeout 300 python3 tests/dev/soak_assoc.py 60 143 2>&1 | tail -1)"
echo "default routing, 45 s, seed 144: $(timeout 300 python3 tests/dev/soak.py 45 144 2>&1 | tail -1)"
} > $out/soak5.log 2>&1
cat $out/soak5.log
bash tools/prof.sh r04f_c5 c5 > $out/prof_c5.log 2>&1
tail -3 $out/prof_c5.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err
tail -c 300 $out/bench_line.json
timeout 600 python3 tests/dev/crossover.py $out/crossover.json > $out/crossover.log 2>&1
tail -5 $out/crossover.log
