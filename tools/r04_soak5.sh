#!/bin/bash
# last soak of the round's last build (commit 5e14a02): the association path once more, longer
out=$(pwd)/gpurun_out/r04_soak5
mkdir -p $out
{
echo "# the round's last build once more, longer (commit 5e14a02)"
echo "exhaustive kernel + bounded walk against the enumeration kernels (SOAK_BNB=1), 240 s, seed 151: $(SOAK_BNB=1 timeout 500 python3 tests/dev/soak_tiny.py 240 151 2>&1 | tail -1)"
echo "association path against the checker, 120 s, seed 152: $(timeout 300 python3 tests/dev/soak_assoc.py 120 152 2>&1 | tail -1)"
echo "association path on the enumeration kernels only (KBEST_NO_TINY KBEST_NO_BNB) against the checker, 100 s, seed 153: $(KBEST_NO_TINY=1 KBEST_NO_BNB=1 timeout 300 python3 tests/dev/soak_assoc.py 100 153 2>&1 | tail -1)"
} > $out/soak.log 2>&1
cat $out/soak.log
