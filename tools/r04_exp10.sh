#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp10
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra --no-host"
km() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % j['kernel_ms'])"; }
for rep in 1 2 3; do
for cfg in c2 c4 c3; do
  echo "$cfg r03: $(KBEST_LIB=libkbest_amd_r03.so timeout 200 $B --config $cfg 2>/dev/null | km)  now: $(timeout 200 $B --config $cfg 2>/dev/null | km)" | tee -a $out/ab.txt
done
done
timeout 1800 python3 -X faulthandler -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/pytest.txt 2>&1
grep -n "passed\|failed\|Fatal\|File \"/.*tests/" $out/pytest.txt | head
