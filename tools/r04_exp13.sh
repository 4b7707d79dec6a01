#!/bin/bash
# C5 (1 000 streamed frames through the small-problem kernel): waves per frame, phase stamps
out=$(pwd)/gpurun_out/r04_exp13
mkdir -p $out
for nw in 2 4 8 16; do
  ( KBEST_SMALL_NW=$nw timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_nw$nw.txt 2>&1
  echo "NW=$nw: $(grep -o '"kernel_ms": [0-9.]*' $out/bench_c5_nw$nw.txt | head -1)"
done
for nw in 2 4 8; do
  ( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=$nw timeout 200 python3 tools/phase_profile.py c5 1000 ) > $out/phase_c5_nw$nw.txt 2>&1
  cat $out/phase_c5_nw$nw.txt
done
