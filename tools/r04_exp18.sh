#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp18
mkdir -p $out
for lib in libkbest_amd_occ6.so libkbest_amd.so; do
for nw in 5 8 10 12 16; do
  ( KBEST_LIB=$lib KBEST_SMALL_NW=$nw timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_${lib}_nw$nw.txt 2>&1
  echo "$lib NW=$nw: $(grep -o '"kernel_ms": [0-9.]*' $out/bench_c5_${lib}_nw$nw.txt | head -1)"
done
done
