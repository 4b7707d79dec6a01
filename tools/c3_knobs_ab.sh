run() { env "$@" timeout 200 python bench.py --config c3 --steps 5 --warmup 2 --no-cpu --no-extra --no-host 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'kernel_ms', round(j['kernel_ms'],3), 'relays', j['launch']['relay_launches'])"; }
for r in 1 2; do
run A=1
run KBEST_NWAVES=8
run KBEST_NWAVES=8 KBEST_SPEC=6
run KBEST_RELAY=4
run KBEST_RELAY=2
run KBEST_OPT_RHO0=0.8
run KBEST_OPT_RHO0=0.9
run KBEST_SPEC=3
done
