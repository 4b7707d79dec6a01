#!/bin/bash
# tools/ab.sh libA.so libB.so ... -- on the GPU box: kernel time of in-tree builds, same inputs (default shape and NW=8 at B=768, B=256)
for lib in "$@"; do
  for cfg in "1024:" "768:8" "256:8"; do
    b=${cfg%%:*}; nw=${cfg##*:}
    r=$(KBEST_LIB=$lib KBEST_NWAVES=$nw timeout 200 python bench.py --steps 10 --warmup 2 --batch $b --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.3f ms' % j['kernel_ms'])")
    echo "$lib B=$b NW=${nw:-auto}: $r"
  done
done
