#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp14
mkdir -p $out
for nw in 4 8; do
  ( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=$nw timeout 200 python3 tests/dev/c5_dist.py 1000 ) 2>&1 | grep -v amdgpu.ids | tee $out/dist_nw$nw.txt
done
( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=4 timeout 200 python3 tests/dev/c5_dist.py 250 ) 2>&1 | grep -v amdgpu.ids | tee $out/dist_nw4_250.txt
( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=4 timeout 200 python3 tests/dev/c5_dist.py 2000 ) 2>&1 | grep -v amdgpu.ids | tee $out/dist_nw4_2000.txt
