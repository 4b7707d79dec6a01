#!/bin/bash
# tools/prof.sh <tag> [config] -- on the GPU box: kernel-trace stats + PMC passes of one bench workload (default c4).
# Summaries land in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
tag=${1:-r02}
cfg=${2:-c4}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
[ -z "$GRAFT_REPO_ROOT" ] && out=$(pwd)/gpurun_out/prof_$tag
mkdir -p $out
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
extra=""
[ "$cfg" = "c5" ] && extra="--kernel-only"
B="python3 $repo/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu --no-extra --no-host $extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $B > $out/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -o pmc -- $B > $out/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq2 -o pmc -- $B > $out/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- $B > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- $B > $out/pmc_write.log 2>&1
cd $repo
python3 tools/prof_summary.py $out 5 $cfg > $out/summary.txt 2>&1
cat $out/summary.txt
