#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp23
mkdir -p $out
S="200 256 300 400 2000 3000 4000 8000"
for nw in 4 8 16; do
  KBEST_SMALL_NW=$nw timeout 300 python3 tests/dev/c5_sweep.py $S 2>&1 | grep "NW="
done
( timeout 300 python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5.txt 2>&1
python3 - <<PY
import json
l=[x for x in open("$out/bench_c5.txt") if x.startswith("{")][-1]
d=json.loads(l)
print("kernel_ms", round(d["kernel_ms"],4), "host", round(d["host_inclusive_batched"]["ms"],3), "one", round(d["one_frame_per_call"]["us_mean"],1), "floor", round(d["one_frame_per_call_floor"]["us_mean"],1), "small", [round(e["us_mean"],1) for e in d["one_frame_per_call_small"]])
PY
