#!/bin/bash
# tools/small_iter.sh -- on the GPU box: quick parity subset + phase profile of the small-problem kernel (dev loop)
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_vectors or dense_configs or random_shapes" 2>&1 | tail -3
KBEST_LIB=libkbest_amd_prof.so python tools/phase_profile.py c5 1 2>/dev/null
KBEST_FORCE_SMALL=1 KBEST_LIB=libkbest_amd_prof.so python tools/phase_profile.py c3 4096 2>/dev/null | grep -E "kernel|dijkstra|rounds|wait|filter|finish|merge|select"
