#!/bin/bash
# same-box A/B of a kbest_create knob on the host-inclusive C4 time: tools/ab_host.sh VAR "v1 v2"   (three interleaved rounds)
var=$1; vals=$2
out=$(pwd)/gpurun_out/ab_host
mkdir -p $out
for r in 1 2 3; do
  for v in $vals; do
    ( env $var=$v timeout 300 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu --no-extra ) > $out/b.txt 2>&1
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b.txt") if l.startswith("{")][-1])
h=d["value_host_inclusive"]
print("round $r $var=$v: pageable", round(h["pageable_ms"],3), "registered", round(h.get("registered_ms") or 0,3), "int8", round(h.get("ms_int8_tables") or 0,3), "kernel", round(d["ms_per_step"],3))
PY
  done
done
