#!/bin/bash
# tools/prof2.sh <tag> <batch> -- issue / cache counters of the k-best kernel at a given batch size
tag=${1:-x}; batch=${2:-768}
repo=$(pwd); out=$repo/gpurun_out/prof2_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $repo/bench.py --steps 5 --warmup 1 --no-cpu --batch $batch"
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_sq -o pmc -- $B > $out/a.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq2 -o pmc -- $B > $out/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LEVEL_WAVES --output-format csv -d $out/pmc_fetch -o pmc -- $B > $out/c.log 2>&1
cd $repo; python3 tools/prof_summary.py $out | grep -v "^==.*kernel dur" 
