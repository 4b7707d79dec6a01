#!/bin/bash
# same-box A/B of the C5 kernel time: tools/ab_c5.sh libA.so libB.so ...   (three interleaved rounds, kernel-only bench)
out=$(pwd)/gpurun_out/ab_c5
mkdir -p $out
for r in 1 2 3; do
  for lib in "$@"; do
    ( KBEST_LIB=$lib timeout 200 python3 bench.py --config c5 --kernel-only --steps 20 --warmup 3 --no-cpu --no-extra ) > $out/b.txt 2>&1
    echo "round $r $lib: $(grep -o '"kernel_ms": [0-9.]*' $out/b.txt | head -1)"
  done
done
