#!/bin/bash
# same-box A/B of a kbest_create knob: tools/ab_env.sh VAR "v1 v2 ..." "cfg1 cfg2 ..."   (three interleaved rounds of the kernel-only bench)
var=$1; vals=$2; cfgs=${3:-c4}
out=$(pwd)/gpurun_out/ab_env
mkdir -p $out
for r in 1 2 3; do
  for c in $cfgs; do
    for v in $vals; do
      ( env $var=$v timeout 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu --no-extra --no-host ) > $out/b.txt 2>&1
      echo "round $r $c $var=$v: $(grep -o '"ms_per_step": [0-9.]*' $out/b.txt | head -1) $(grep -o '"parity_vs_gpu": [a-z]*' $out/b.txt | head -1)"
    done
  done
done
