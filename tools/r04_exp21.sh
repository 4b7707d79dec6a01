#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp21
mkdir -p $out
run() { ( env "$@" timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/b.txt 2>&1; echo "$*: $(grep -o '"kernel_ms": [0-9.]*' $out/b.txt | head -1)"; }
run KBEST_SMALL_FPW=1
run KBEST_SMALL_FPW=4
run KBEST_SMALL_FPW=2 KBEST_SMALL_NW=5
run KBEST_LIB=libkbest_amd_occ6.so KBEST_SMALL_FPW=2 KBEST_SMALL_NW=5
run KBEST_LIB=libkbest_amd_occ6.so KBEST_SMALL_NW=5
run KBEST_LIB=libkbest_amd_occ6.so KBEST_SMALL_NW=16
run KBEST_SMALL_NW=16
