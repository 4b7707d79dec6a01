#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp24
mkdir -p $out
timeout 900 python3 -m pytest tests -q -m gpu -k "assoc or weight or prob or frames or config5" 2>&1 | grep -E "passed|failed|rror" | tail -5
( timeout 300 python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5.txt 2>&1
python3 - <<PY
import json
l=[x for x in open("$out/bench_c5.txt") if x.startswith("{")][-1]
d=json.loads(l)
print("kernel_ms", round(d["kernel_ms"],4), "host", round(d["host_inclusive_batched"]["ms"],3), "one", round(d["one_frame_per_call"]["us_mean"],1), "floor", round(d["one_frame_per_call_floor"]["us_mean"],1), "small", [round(e["us_mean"],1) for e in d["one_frame_per_call_small"]])
PY
