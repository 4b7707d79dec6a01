#!/bin/bash
var=$1; vals=$2
out=$(pwd)/gpurun_out/ab_two
mkdir -p $out
for r in 1 2; do
  for v in $vals; do
    ( env $var=$v timeout 400 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu ) > $out/b.txt 2>&1
    python3 - <<PY
import json
d=json.loads([l for l in open("$out/b.txt") if l.startswith("{")][-1])
c=d["configs"]; h=d["value_host_inclusive"]
print("round $r $var=$v: two_in_flight", round(c["c4_two_batches_in_flight"]["ms_per_batch"],3), "multi1", round(c["c4_multi_entry_1dev"]["ms"],3), "multi8", round(c["c4_multi_entry_8dev"]["ms"],3), "pageable", round(h["pageable_ms"],3), "kernel", round(d["ms_per_step"],3))
PY
  done
done
