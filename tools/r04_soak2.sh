#!/bin/bash
# second soak of the round's last build: the kernels that changed after the first (small-problem kernel shapes, exhaustive kernel)
out=$(pwd)/gpurun_out/r04_soak2
mkdir -p $out
{
echo "# after the last kernel changes of the round (small-problem kernel at six waves per SIMD, 8 waves per problem up to 6 per CU; exhaustive kernel behind both association entries)"
echo "small-problem kernel forced (KBEST_FORCE_SMALL), 240 s, seed 61: $(KBEST_FORCE_SMALL=1 timeout 400 python3 tests/dev/soak.py 240 61 2>&1 | tail -1)"
echo "small-problem kernel forced, 4 waves per problem, 120 s, seed 62: $(KBEST_FORCE_SMALL=1 KBEST_SMALL_NW=4 timeout 300 python3 tests/dev/soak.py 120 62 2>&1 | tail -1)"
echo "default routing, 180 s, seed 63: $(timeout 400 python3 tests/dev/soak.py 180 63 2>&1 | tail -1)"
echo "association path, 240 s, seed 64: $(timeout 400 python3 tests/dev/soak_assoc.py 240 64 2>&1 | tail -1)"
echo "exhaustive kernel against the enumeration kernels, 240 s, seed 65: $(timeout 400 python3 tests/dev/soak_tiny.py 240 65 2>&1 | tail -1)"
} > $out/soak2.log 2>&1
cat $out/soak2.log
