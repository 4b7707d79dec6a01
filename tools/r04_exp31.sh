timeout 600 python3 -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "bounded or frame_sized or fused or assoc" 2>&1 | tail -2
SOAK_BNB=1 timeout 200 python3 tests/dev/soak_tiny.py 45 131 2>&1 | tail -1
timeout 200 python3 tests/dev/soak_assoc.py 30 132 2>&1 | tail -1
timeout 300 python3 tests/dev/bnb_diag.py 2>&1 | grep -v "handed back"
timeout 300 python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['configs']['c5'] if 'configs' in d else d
        print('kernel_ms',c['kernel_ms'],'one frame',c['one_frame_per_call']['us_mean'],c['one_frame_per_call']['us_p95'],'batched host',c['host_inclusive_batched']['ms'])
"
