#!/bin/bash
# tools/occ_scan.sh -- kernel time vs matrices per CU (NW=8): tells latency-bound (flat) from issue-bound (linear)
for b in 256 512 768 1536; do
  r=$(KBEST_NWAVES=8 timeout 200 python bench.py --steps 10 --warmup 2 --batch $b --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms' % j['kernel_ms'])")
  echo "NW=8 B=$b: $r"
done
for b in 256 512 1024; do
  r=$(KBEST_NWAVES=12 KBEST_SPEC=8 timeout 200 python bench.py --steps 10 --warmup 2 --batch $b --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms' % j['kernel_ms'])")
  echo "NW=12 B=$b: $r"
done
