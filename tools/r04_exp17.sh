#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp17
mkdir -p $out
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or random_shapes" 2>&1 | tail -1
KBEST_LIB=libkbest_amd_occ8.so timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or random_shapes" 2>&1 | tail -1
for lib in libkbest_amd.so libkbest_amd_occ6.so libkbest_amd_occ8.so; do
for nw in 4 5 6 8; do
  ( KBEST_LIB=$lib KBEST_SMALL_NW=$nw timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_${lib}_nw$nw.txt 2>&1
  echo "$lib NW=$nw: $(grep -o '"kernel_ms": [0-9.]*' $out/bench_c5_${lib}_nw$nw.txt | head -1)"
done
done
