#!/bin/bash
# tools/r04_exp1.sh -- on the GPU box: where the round starts.  Phase stamps (PROFILE build) of every BASELINE config's
# kernel and the launch-shape knobs that are cheap to scan.  Output: gpurun_out/r04_exp1/*.txt
out=$(pwd)/gpurun_out/r04_exp1
mkdir -p $out
P="python3 tools/phase_profile.py"
export KBEST_LIB=libkbest_amd_prof.so
( KBEST_NWAVES=4 KBEST_SPEC=4 timeout 200 $P c3 ) > $out/phase_c3_nw4.txt 2>&1
( KBEST_NWAVES=8 KBEST_SPEC=8 timeout 200 $P c3 ) > $out/phase_c3_nw8.txt 2>&1
( KBEST_FORCE_LANE=1 KBEST_LANE_NW=2 timeout 200 $P c3 ) > $out/phase_c3_lane2.txt 2>&1
( KBEST_FORCE_LANE=1 KBEST_LANE_NW=4 timeout 200 $P c2 ) > $out/phase_c2_lane4.txt 2>&1
( KBEST_NWAVES=12 KBEST_SPEC=12 timeout 200 $P c4 ) > $out/phase_c4.txt 2>&1
for nw in 4 8 16; do
  ( KBEST_SMALL_NW=$nw timeout 200 $P c5 1000 ) > $out/phase_c5_nw$nw.txt 2>&1
done
( timeout 200 $P c5 1 ) > $out/phase_c5_one.txt 2>&1
unset KBEST_LIB
for nw in 4 8 16; do
  ( KBEST_SMALL_NW=$nw timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra --no-host ) > $out/bench_c5_nw$nw.txt 2>&1
done
( timeout 600 python3 bench.py --steps 10 --warmup 2 ) > $out/bench_default.txt 2>&1
tail -n 30 $out/phase_c3_nw4.txt $out/phase_c5_nw4.txt
for f in $out/bench_c5_nw*.txt; do echo $f; tail -1 $f | cut -c1-300; done
