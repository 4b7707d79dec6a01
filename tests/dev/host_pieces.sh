for i in 1 2; do
for cfg in "KBEST_PIECES=4" "KBEST_PIECES=2" "KBEST_PIECES=1"; do
  echo "== $cfg: $(env $cfg HP_ONE=1 timeout 200 python3 tests/dev/host_pieces.py 2>&1 | grep -v amdgpu | tail -1)"
done; done
