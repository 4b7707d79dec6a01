#!/bin/bash
# tools/shape_scan.sh [B] -- kernel time over (waves per matrix, candidates split per round)
b=${1:-1024}
for cfg in "8 4" "8 5" "8 6" "8 7" "12 5" "12 6" "12 7" "12 8" "16 8"; do
  set -- $cfg
  r=$(KBEST_NWAVES=$1 KBEST_SPEC=$2 timeout 200 python bench.py --steps 8 --warmup 2 --batch $b --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms' % j['kernel_ms'])")
  echo "B=$b NW=$1 spec=$2: $r"
done
