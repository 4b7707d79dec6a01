#!/bin/bash
# on the GPU box: tests, a short soak and the one-frame timing of the exhaustive kernel (dev loop)
timeout 800 python3 -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "handful or exhaustive" 2>&1 | tail -2
timeout 200 python3 tests/dev/soak_tiny.py 60 ${1:-9} | tail -2
timeout 600 python3 tests/dev/tiny_time.py 2>&1 | grep nL=
