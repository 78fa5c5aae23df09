python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --config ${CFG:-c4} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'], j['stats'])"; }
run default
FTKX_TWO_LEVEL=0 run onelevel
CFG=c3 run c3
CFG=c2 run c2
CFG=c2 FTKX_TWO_LEVEL=0 run c2_onelevel
CFG=c1 run c1
