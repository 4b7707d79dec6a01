#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp16
mkdir -p $out
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or random_shapes" 2>&1 | tail -2
for nw in 3 4 5 6; do
  ( KBEST_SMALL_NW=$nw timeout 200 python3 bench.py --config c5 --kernel-only --steps 10 --warmup 2 --no-cpu --no-extra ) > $out/bench_c5_nw$nw.txt 2>&1
  echo "NW=$nw: $(grep -o '"kernel_ms": [0-9.]*' $out/bench_c5_nw$nw.txt | head -1)"
done
for nw in 5; do
  ( KBEST_LIB=libkbest_amd_prof.so KBEST_SMALL_NW=$nw timeout 200 python3 tests/dev/c5_dist.py 1000 ) 2>&1 | grep -v "amdgpu.ids\|Warning\|stddev" | tee $out/dist_nw$nw.txt
done
KBEST_SMALL_NW=5 timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_weights.py -x -q -m gpu 2>&1 | tail -2
