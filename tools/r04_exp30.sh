#!/bin/bash
# in-pass bound lowering of the bounded walk: parity first, then same-box A/B of the start factor
mkdir -p gpurun_out/exp30
timeout 600 python3 -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "bounded or frame_sized or fused or assoc" > gpurun_out/exp30/tests.txt 2>&1; tail -3 gpurun_out/exp30/tests.txt
P=$(pwd)/probabilisticsemslam_amd
for lib in base amd_a amd_b amd_c amd_d amd_e; do echo "$lib: $(KBEST_LIB=$P/libkbest_$lib.so timeout 200 python3 tests/dev/bnb_diag.py 2>&1 | grep -A8 'F=1000' | tr '\n' ' ')"; done
bash tools/ab_c5.sh $P/libkbest_base.so $P/libkbest_amd_a.so $P/libkbest_amd_b.so $P/libkbest_amd_c.so $P/libkbest_amd_d.so $P/libkbest_amd_e.so
