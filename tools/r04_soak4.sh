#!/bin/bash
# last soak of the round's last build: the association path once more, longer
out=$(pwd)/gpurun_out/r04_soak4
mkdir -p $out
{
echo "# the round's last build (bounded walk with its rank loop unrolled; commit f871cba and later)"
echo "exhaustive kernel + bounded walk against the enumeration kernels (SOAK_BNB=1), 300 s, seed 101: $(SOAK_BNB=1 timeout 600 python3 tests/dev/soak_tiny.py 300 101 2>&1 | tail -1)"
echo "association path against the checker, 240 s, seed 102: $(timeout 500 python3 tests/dev/soak_assoc.py 240 102 2>&1 | tail -1)"
echo "default routing, 180 s, seed 103: $(timeout 400 python3 tests/dev/soak.py 180 103 2>&1 | tail -1)"
} > $out/soak4.log 2>&1
cat $out/soak4.log
