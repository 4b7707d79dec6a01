#!/bin/bash
# tools/gpu_check.sh [tag] [pytest-args] -- on the GPU box: GPU parity tests, then the default bench line.
tag=${1:-check}
shift
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_$tag.log
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$tag.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_$tag.log | python3 -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('value %.4g %s  ms/step %.3f  kernel_ms %.3f  roofline.frac %.4f  parity_self %s  cpu %s  speedup %s' % (j['value'], j['unit'], j['ms_per_step'], j['kernel_ms'], j['roofline']['frac'], j['parity_prune_vs_noprune'], j.get('cpu_baseline', {}).get('value'), j.get('speedup_vs_cpu_1core')))
except Exception as e:
    print('bench parse failed', e)
"
