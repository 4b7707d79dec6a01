#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp19
mkdir -p $out
for f in 4 2; do
  echo "== FPW=$f, 4 waves per problem: parity"
  KBEST_SMALL_FPW=$f KBEST_SMALL_NW=4 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
done
for f in 1 2 4; do
  ( KBEST_SMALL_FPW=$f timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_fpw$f.txt 2>&1
  echo "FPW=$f: $(grep -o '"kernel_ms": [0-9.]*' $out/bench_c5_fpw$f.txt | head -1)"
done
