#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp15
mkdir -p $out
for nw in 16 8; do
( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=$nw timeout 200 python3 tools/phase_profile.py c5 1 ) 2>&1 | grep -v amdgpu.ids | tee $out/phase_c5_b1_nw$nw.txt
done
( KBEST_LIB=libkbest_amd_prof.so timeout 200 python3 tools/phase_profile.py c2 1024 ) 2>&1 | grep -v amdgpu.ids | tee $out/phase_c2.txt
