#!/bin/bash
# same-box A/B of the association path: tools/ab_c5_full.sh libA.so libB.so ...  (batched kernel time and the one-frame-per-call time)
out=$(pwd)/gpurun_out/ab_c5
mkdir -p $out
for r in 1 2; do
  for lib in "$@"; do
    ( KBEST_LIB=$lib timeout 300 python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu ) > $out/b.txt 2>&1
    python3 - <<PY
import json
d=None
for l in open("$out/b.txt"):
    if l.startswith("{"): d=json.loads(l)
c=d["configs"]["c5"] if "configs" in d and "c5" in d["configs"] else d
o=c.get("one_frame_per_call",{})
print("round $r $(basename $lib): kernel_ms", round(c.get("kernel_ms",d.get("kernel_ms",0)),4), "one frame us mean", round(o.get("us_mean",0),1), "median", round(o.get("us_median",0),1), "p95", round(o.get("us_p95",0),1))
PY
  done
done
