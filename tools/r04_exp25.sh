#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp25
mkdir -p $out
for t in tiny notiny; do
  if [ $t = notiny ]; then export KBEST_NO_TINY=1; else unset KBEST_NO_TINY; fi
  ( timeout 300 python3 bench.py --config c5 --steps 10 --warmup 2 --no-extra ) > $out/bench_c5_$t.txt 2>&1
  python3 - <<PY
import json
l=[x for x in open("$out/bench_c5_$t.txt") if x.startswith("{")][-1]
d=json.loads(l)
print("$t", "kernel_ms", round(d["kernel_ms"],4), "one", round(d["one_frame_per_call"]["us_mean"],1), "floor", round(d["one_frame_per_call_floor"]["us_mean"],1))
for e in d["one_frame_per_call_small"]: print("   ", {k:(round(v,1) if isinstance(v,float) else v) for k,v in e.items()})
PY
done
unset KBEST_NO_TINY
ls tests/dev/crossover.py && timeout 600 python3 tests/dev/crossover.py > $out/crossover.txt 2>&1; tail -30 $out/crossover.txt
