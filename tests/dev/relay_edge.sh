# tests/dev/relay_edge.sh [config] -- batches just above one generation of resident workgroups: plain, the plan, and forced piece counts / cuts
cfg=${1:-c4}
for b in ${BATCHES:-513 530 600 700 768}; do
  echo "== $cfg B=$b"
  echo "plain          $(KBEST_RELAY=0 timeout 100 python3 tests/dev/relay_one.py $cfg $b 2>&1 | tail -1 | sed 's/.*min/min/; s/gsum.*//')"
  echo "plan           $(timeout 100 python3 tests/dev/relay_one.py $cfg $b 2>&1 | tail -1 | sed 's/.*min/min/; s/gsum.*//')"
  for pfs in "3 384 384" "4 256 256" "5 205 205" "6 170 170" "8 128 128" "4 384 213"; do
    set -- $pfs
    echo "P=$1 F=$2 S=$3  $(KBEST_RELAY=$1 KBEST_RELAY_FIRST=$2 KBEST_RELAY_STEP=$3 timeout 100 python3 tests/dev/relay_one.py $cfg $b 2>&1 | tail -1 | sed 's/.*min/min/; s/gsum.*//')"
  done
done
