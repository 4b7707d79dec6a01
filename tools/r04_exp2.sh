#!/bin/bash
# tools/r04_exp2.sh -- on the GPU box: parity of the optimistic-bound build, then A/B against KBEST_NO_OPT and a scan of the quantile
out=$(pwd)/gpurun_out/r04_exp2
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "golden or dense_configs or random_shapes or ties or every_launch_shape or column_order or full_size or soak or a_priori or subtree or device_merge" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra --no-host"
km() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % j['kernel_ms'])"; }
for cfg in c4 c3 c2; do
  echo "$cfg no-opt: $(KBEST_NO_OPT=1 timeout 200 $B --config $cfg 2>/dev/null | km)  default: $(timeout 200 $B --config $cfg 2>/dev/null | km)" | tee -a $out/ab.txt
done
for r in "0.4 1.0 1.0" "0.55 1.0 1.0" "0.7 1.0 1.0" "0.55 1.0 0.5" "0.7 0.7 1.0" "0.85 0.85 1.0"; do
  set -- $r
  echo "c4 rho0=$1 rho1=$2 phi=$3: $(KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c4 2>/dev/null | km)  noT0: $(KBEST_NO_T0=1 KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c4 2>/dev/null | km)" | tee -a $out/scan.txt
done
for r in "0.7 0.7 1.0" "0.8 0.8 1.0" "0.85 0.85 1.0" "0.9 0.9 1.0" "0.7 0.95 0.5" "0.55 1.0 1.0"; do
  set -- $r
  echo "c3 rho0=$1 rho1=$2 phi=$3: $(KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c3 2>/dev/null | km)  nw8: $(KBEST_NWAVES=8 KBEST_SPEC=8 KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c3 2>/dev/null | km)" | tee -a $out/scan.txt
done
export KBEST_LIB=libkbest_amd_prof.so
( KBEST_NWAVES=4 KBEST_SPEC=4 timeout 200 python3 tools/phase_profile.py c3 ) > $out/phase_c3.txt 2>&1
( KBEST_NWAVES=12 KBEST_SPEC=12 timeout 200 python3 tools/phase_profile.py c4 ) > $out/phase_c4.txt 2>&1
grep -E "children|steps|rounds|kernel" $out/phase_c3.txt $out/phase_c4.txt
