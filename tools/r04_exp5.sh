#!/bin/bash
# tools/r04_exp5.sh -- on the GPU box: optimistic bounds, third version: A/B, quantile scan, phase stamps, tests
out=$(pwd)/gpurun_out/r04_exp5
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra --no-host"
km() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % j['kernel_ms'])"; }
for rep in 1 2; do
for cfg in c4 c3; do
  echo "$cfg no-opt: $(KBEST_NO_OPT=1 timeout 200 $B --config $cfg 2>/dev/null | km)  default: $(timeout 200 $B --config $cfg 2>/dev/null | km)" | tee -a $out/ab.txt
done
done
for r in "0.7 1.0 1.0" "0.8 1.0 1.0" "0.9 1.0 1.0"; do
  set -- $r
  echo "c4 rho0=$1 rho1=$2 phi=$3: $(KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c4 2>/dev/null | km)  noT0: $(KBEST_NO_T0=1 KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c4 2>/dev/null | km)" | tee -a $out/scan.txt
done
for r in "0.75 0.75 1.0" "0.8 0.8 1.0" "0.85 0.85 1.0" "0.75 0.95 0.5" "0.8 0.95 0.5"; do
  set -- $r
  echo "c3 rho0=$1 rho1=$2 phi=$3: $(KBEST_OPT_RHO0=$1 KBEST_OPT_RHO1=$2 KBEST_OPT_PHI=$3 timeout 200 $B --config c3 2>/dev/null | km)" | tee -a $out/scan.txt
done
export KBEST_LIB=libkbest_amd_prof.so
( KBEST_NWAVES=4 KBEST_SPEC=4 timeout 200 python3 tools/phase_profile.py c3 ) > $out/phase_c3_opt.txt 2>&1
unset KBEST_LIB
timeout 900 python3 -m pytest tests/test_gpu_round4.py -q -m gpu > $out/pytest4.txt 2>&1
tail -5 $out/pytest4.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
