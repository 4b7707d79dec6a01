#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp11
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra --no-host"
km() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % j['kernel_ms'])"; }
for rep in 1 2 3; do
  echo "c3 now: $(timeout 200 $B --config c3 2>/dev/null | km)  occ7: $(KBEST_LIB=libkbest_amd_occ7.so timeout 200 $B --config c3 2>/dev/null | km)" | tee -a $out/ab.txt
done
for b in 2048 8192; do
  echo "32x32 B=$b now: $(timeout 200 $B --config c3 --batch $b 2>/dev/null | km)  occ7: $(KBEST_LIB=libkbest_amd_occ7.so timeout 200 $B --config c3 --batch $b 2>/dev/null | km)  r03: $(KBEST_LIB=libkbest_amd_r03.so timeout 200 $B --config c3 --batch $b 2>/dev/null | km)" | tee -a $out/ab.txt
done
