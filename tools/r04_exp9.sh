#!/bin/bash
# tools/r04_exp9.sh -- on the GPU box: why is C4 slower than with round 3's library?  SQ counters of both; the blocked closure's effect
out=$(pwd)/gpurun_out/r04_exp9
mkdir -p $out
repo=$(pwd)
B="python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra --no-host"
km() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % j['kernel_ms'])"; }
for rep in 1 2; do
for cfg in c4 c3; do
  echo "$cfg r03: $(KBEST_LIB=libkbest_amd_r03.so timeout 200 $B --config $cfg 2>/dev/null | km)  now: $(timeout 200 $B --config $cfg 2>/dev/null | km)  now, no reorder: $(KBEST_NO_REORDER=1 timeout 200 $B --config $cfg 2>/dev/null | km)  r03, no reorder: $(KBEST_LIB=libkbest_amd_r03.so KBEST_NO_REORDER=1 timeout 200 $B --config $cfg 2>/dev/null | km)" | tee -a $out/ab.txt
done
done
export KBEST_LIB=libkbest_amd_prof.so
( KBEST_NWAVES=12 KBEST_SPEC=12 timeout 200 python3 tools/phase_profile.py c4 ) > $out/phase_c4.txt 2>&1
unset KBEST_LIB
grep -E "children|steps|rounds|kernel|setup" $out/phase_c4.txt
cd /tmp && export TMPDIR=/tmp
BB="python3 $repo/bench.py --config c4 --steps 3 --warmup 1 --no-cpu --no-extra --no-host"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_now -o pmc -- $BB > $out/pmc_now.log 2>&1
export KBEST_LIB=libkbest_amd_r03.so
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_r03 -o pmc -- $BB > $out/pmc_r03.log 2>&1
unset KBEST_LIB
cd $repo
for d in pmc_now pmc_r03; do
  f=$(find $out/$d -name "*counter_collection.csv" | head -1)
  echo "== $d"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "kbest_kernel" in r.get("Kernel_Name", ""):
        acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(dict)
for (d, c), v in acc.items():
    per[d][c] = sum(v)
ds = sorted(per, key=lambda x: int(x))[-3:]
for c in sorted(per[ds[0]]):
    print(c, sum(per[d][c] for d in ds) / len(ds))
PY
done
timeout 600 python3 -X faulthandler -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "several_workgroups" -p no:cacheprovider > $out/pytest_split.txt 2>&1
tail -30 $out/pytest_split.txt | cut -c1-200
