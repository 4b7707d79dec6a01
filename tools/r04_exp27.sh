#!/bin/bash
timeout 800 python3 -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "handful or exhaustive" 2>&1 | tail -3
timeout 600 python3 tests/dev/tiny_time.py 2>&1 | grep nL=
