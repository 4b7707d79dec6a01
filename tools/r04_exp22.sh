#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp22
mkdir -p $out
S="300 512 600 768 900 1000 1024 1100 1300 2000"
for nw in rule 4 5 8; do
  if [ $nw = rule ]; then unset KBEST_SMALL_NW; else export KBEST_SMALL_NW=$nw; fi
  timeout 300 python3 tests/dev/c5_sweep.py $S 2>&1 | grep "NW="
done
unset KBEST_SMALL_NW
for nw in 4 8; do
  KBEST_LIB=libkbest_amd_occ5.so KBEST_SMALL_NW=$nw timeout 300 python3 tests/dev/c5_sweep.py $S 2>&1 | grep "NW=" | sed 's/^/occ5 /'
done
for lib in libkbest_amd_occ5.so libkbest_amd.so; do
  ( KBEST_LIB=$lib timeout 300 python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_$lib.txt 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/bench_c5_$lib.txt") if x.startswith("{")][-1]
d=json.loads(l)
print("$lib", "kernel_ms", round(d["kernel_ms"],4), "host", round(d["host_inclusive_batched"]["ms"],3), "one", round(d["one_frame_per_call"]["us_mean"],1), "floor", round(d["one_frame_per_call_floor"]["us_mean"],1), "small", [round(e["us_mean"],1) for e in d["one_frame_per_call_small"]])
PY
done
