#!/bin/bash
out=$(pwd)/gpurun_out/r04_exp26
mkdir -p $out
timeout 800 python3 -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "handful or exhaustive" 2>&1 | tail -3
timeout 600 python3 tests/dev/crossover.py > $out/crossover.txt 2>&1; tail -6 $out/crossover.txt
