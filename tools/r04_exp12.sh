#!/bin/bash
# what would optimistic bounds be worth on the KITTI-like frames?  The 64-row kernel (which has them) on the conditioned C5 frames:
# children / steps / completions with and without, PROFILE build
out=$(pwd)/gpurun_out/r04_exp12
mkdir -p $out
export KBEST_LIB=libkbest_amd_prof.so KBEST_NO_SMALL=1 KBEST_NWAVES=4 KBEST_SPEC=4 KBEST_OPT_RHO0=0.85
for r in off 0.85 0.7 0.55; do
  if [ $r = off ]; then export KBEST_NO_OPT=1; else unset KBEST_NO_OPT; export KBEST_OPT_RHO0=$r; fi
  ( timeout 200 python3 tools/phase_profile.py c5 1000 ) > $out/phase_c5_engine_$r.txt 2>&1
  echo "rho $r: $(grep -E 'kernel [0-9.]+ ms' $out/phase_c5_engine_$r.txt | cut -c1-80) $(grep -E '\[ 4\]|\[ 5\]|\[ 6\]|\[ 7\]' $out/phase_c5_engine_$r.txt | tr -s ' ' | tr '\n' ';')"
done
