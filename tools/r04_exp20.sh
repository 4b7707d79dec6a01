#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp20
mkdir -p $out
for f in 4; do
  echo "== FPW=$f, 4 waves per problem: parity"
  KBEST_SMALL_FPW=$f KBEST_SMALL_NW=4 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
done
for f in 1 4; do
  ( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_FPW=$f KBEST_SMALL_NW=4 timeout 200 python3 tools/phase_profile.py c5 1000 ) 2>&1 | grep -v amdgpu.ids | tee $out/phase_fpw$f.txt
  ( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_FPW=$f KBEST_SMALL_NW=4 timeout 200 python3 tests/dev/c5_dist.py 1000 ) 2>&1 | grep -v "amdgpu.ids\|Warning\|stddev" | tee $out/dist_fpw$f.txt
done
