#!/bin/bash
# tools/r04_final.sh -- on the GPU box: the evidence of the round: soak (every routing), profiles of every config, bench line, crossover
out=$(pwd)/gpurun_out/r04_final
mkdir -p $out
{
echo "# round 4, final build: tests/dev/soak.py / soak_assoc.py (random shapes, flags, cost structures against the checker)"
echo "default routing, 240 s, seed 41: $(timeout 400 python3 tests/dev/soak.py 240 41 2>&1 | tail -1)"
echo "64-row kernel only (KBEST_NO_SMALL, KBEST_NO_LANE), 4 waves x 4 (optimistic bounds on), 240 s, seed 42: $(KBEST_NO_SMALL=1 KBEST_NO_LANE=1 KBEST_NWAVES=4 KBEST_SPEC=4 timeout 400 python3 tests/dev/soak.py 240 42 2>&1 | tail -1)"
echo "64-row kernel only, 4 waves x 4, aggressive quantile (KBEST_OPT_RHO0=0.4: many tickets), 180 s, seed 43: $(KBEST_NO_SMALL=1 KBEST_NO_LANE=1 KBEST_NWAVES=4 KBEST_SPEC=4 KBEST_OPT_RHO0=0.4 timeout 400 python3 tests/dev/soak.py 180 43 2>&1 | tail -1)"
echo "64-row kernel only, 2 waves x 2, quantile 0.6, 120 s, seed 44: $(KBEST_NO_SMALL=1 KBEST_NO_LANE=1 KBEST_NWAVES=4 KBEST_SPEC=2 KBEST_OPT_RHO0=0.6 timeout 300 python3 tests/dev/soak.py 120 44 2>&1 | tail -1)"
echo "64-row kernel only, 12 waves x 12, 120 s, seed 45: $(KBEST_NO_SMALL=1 KBEST_NO_LANE=1 KBEST_NWAVES=12 KBEST_SPEC=12 timeout 300 python3 tests/dev/soak.py 120 45 2>&1 | tail -1)"
echo "lane-per-child kernel forced, 120 s, seed 46: $(KBEST_FORCE_LANE=1 timeout 300 python3 tests/dev/soak.py 120 46 2>&1 | tail -1)"
echo "general-size kernel forced, 120 s, seed 47: $(KBEST_FORCE_WIDE=1 timeout 300 python3 tests/dev/soak.py 120 47 2>&1 | tail -1)"
echo "rows up to 1 024 (SOAK_BIG=0.5 SOAK_BIGMAX=1024), 240 s, seed 48: $(SOAK_BIG=0.5 SOAK_BIGMAX=1024 timeout 500 python3 tests/dev/soak.py 240 48 2>&1 | tail -1)"
echo "exhaustive kernel against the enumeration kernels (frames of 2-8 measurements through kbest_assoc_probs_batch_f64), 120 s, seed 50: $(timeout 300 python3 tests/dev/soak_tiny.py 120 50 2>&1 | tail -1)"
echo "association path, 120 s, seed 49: $(timeout 300 python3 tests/dev/soak_assoc.py 120 49 2>&1 | tail -1)"
} > $out/soak.log 2>&1
cat $out/soak.log
for c in c4 c3 c2 c5 w128; do
  bash tools/prof.sh r04f_$c $c > $out/prof_$c.log 2>&1
  tail -3 $out/prof_$c.log
done
timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err
tail -c 400 $out/bench_line.json
timeout 600 python3 tests/dev/crossover.py $out/crossover.json > $out/crossover.log 2>&1
tail -5 $out/crossover.log
