#!/bin/bash
# tools/r04_exp4.sh -- on the GPU box: where the optimistic-bound build spends what it saves (C3): phase stamps and SQ counters, with and without
out=$(pwd)/gpurun_out/r04_exp4
mkdir -p $out
repo=$(pwd)
export KBEST_LIB=libkbest_amd_prof.so
( KBEST_NWAVES=4 KBEST_SPEC=4 timeout 200 python3 tools/phase_profile.py c3 ) > $out/phase_c3_opt.txt 2>&1
( KBEST_NO_OPT=1 KBEST_NWAVES=4 KBEST_SPEC=4 timeout 200 python3 tools/phase_profile.py c3 ) > $out/phase_c3_noopt.txt 2>&1
( KBEST_NWAVES=12 KBEST_SPEC=12 KBEST_OPT_RHO0=0.8 timeout 200 python3 tools/phase_profile.py c4 ) > $out/phase_c4_opt.txt 2>&1
( KBEST_NO_OPT=1 KBEST_NWAVES=12 KBEST_SPEC=12 timeout 200 python3 tools/phase_profile.py c4 ) > $out/phase_c4_noopt.txt 2>&1
unset KBEST_LIB
cd /tmp && export TMPDIR=/tmp
B="python3 $repo/bench.py --config c3 --steps 3 --warmup 1 --no-cpu --no-extra --no-host"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_opt -o pmc -- $B > $out/pmc_opt.log 2>&1
export KBEST_NO_OPT=1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_noopt -o pmc -- $B > $out/pmc_noopt.log 2>&1
unset KBEST_NO_OPT
cd $repo
for d in pmc_opt pmc_noopt; do
  f=$(find $out/$d -name "*counter_collection.csv" | head -1)
  echo "== $d $f"
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
timeout 600 python3 -m pytest tests/test_gpu_round4.py -q -m gpu > $out/pytest4.txt 2>&1
tail -5 $out/pytest4.txt
