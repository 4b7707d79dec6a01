#!/bin/bash
# tools/soak.sh <tag> [seconds per leg] [seed] -- on the GPU box: the fixed-seed soaks of the build, one log under gpurun_out/<tag>/
# (copy what should be judged to profiles/<tag>_soak.log).  Legs: the k-best tables of every kernel against the checker
# (tests/dev/soak.py; again with every 64-row launch forced into a relay of three pieces, and big relayed batches against plain ones: tests/dev/relay_stress.py), the association path against the checker -- on the fast kernels and on the enumeration kernels alone
# (tests/dev/soak_assoc.py) --, and the exhaustive kernel + bounded walk against the enumeration kernels, exact ties included
# (tests/dev/soak_tiny.py).
tag=${1:-soak}; secs=${2:-60}; seed=${3:-500}
out=$(pwd)/gpurun_out/$tag
mkdir -p $out
t=$((secs * 3 + 120))
{
echo "# soak of $(git rev-parse --short HEAD 2>/dev/null || echo 'the working tree'), $secs s per leg, seeds $seed.."
echo "k-best tables, every kernel against the checker (KBEST_FLAG_CANONICAL_TIES: the engine's own rule on exact ties, equal gains compared as sets): $(timeout $t python3 tests/dev/soak.py $secs $seed 2>&1 | tail -1)"
for nw in 4 8 12; do
echo "k-best tables, every 64-row launch forced into a relay of three pieces of the $nw-wave shape (KBEST_RELAY=3 KBEST_NWAVES=$nw KBEST_NO_SMALL KBEST_NO_LANE) against the checker: $(KBEST_RELAY=3 KBEST_NWAVES=$nw KBEST_NO_SMALL=1 KBEST_NO_LANE=1 timeout $t python3 tests/dev/soak.py $secs $((seed + 5 + nw)) 2>&1 | tail -1)"
done
echo "k-best tables in the reference's own order (KBEST_FLAG_REFERENCE_ORDER, kbest_exact.hip) against the checker SLOT FOR SLOT -- order of equal gains and col4row on padded columns included: $(SOAK_REFERENCE_ORDER=1 timeout $t python3 tests/dev/soak.py $secs $((seed + 40)) 2>&1 | tail -1)"
echo "k-best tables as the synchronous entry returns them BY DEFAULT (the fast kernels; every problem with an exact tie among its k + 1 best gains again on the reference-order kernel) against the checker: gains and row4col SLOT FOR SLOT on every problem: $(SOAK_REFERENCE_ORDER=2 timeout $t python3 tests/dev/soak.py $secs $((seed + 41)) 2>&1 | tail -1)"
echo "relay launches of batches of several generations (the plan, 2, 5, 8 pieces) against plain launches, every table word: $(timeout $t python3 tests/dev/relay_stress.py $secs $((seed + 30)) 2>&1 | tail -1)"
echo "association path against the checker: $(timeout $t python3 tests/dev/soak_assoc.py $secs $((seed + 1)) 2>&1 | tail -1)"
echo "association path on the enumeration kernels only (KBEST_NO_TINY KBEST_NO_BNB) against the checker: $(KBEST_NO_TINY=1 KBEST_NO_BNB=1 timeout $t python3 tests/dev/soak_assoc.py $secs $((seed + 2)) 2>&1 | tail -1)"
echo "association path on INTEGER costs with kbest_set_reference_order(ctx, 2) (fused kernels; the frames with a tie at slot k again on the reference-order kernel) against the checker's probabilities -- the reference's own choice among a tied level: $(SOAK_ASSOC_REFERENCE=2 timeout $t python3 tests/dev/soak_assoc.py $secs $((seed + 42)) 2>&1 | tail -1)"
echo "exhaustive kernel against the enumeration kernels, exact ties included: $(timeout $t python3 tests/dev/soak_tiny.py $secs $((seed + 3)) 2>&1 | tail -1)"
echo "exhaustive kernel + bounded walk against the enumeration kernels (SOAK_BNB=1), exact ties included: $(SOAK_BNB=1 timeout $t python3 tests/dev/soak_tiny.py $secs $((seed + 4)) 2>&1 | tail -1)"
} > $out/soak.log 2>&1
cat $out/soak.log
