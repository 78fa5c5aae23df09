python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --config ${CFG:-c4} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'])"; }
run default
FTKX_MASK_WPB=2 run wpb2
FTKX_MASK_WPB=3 run wpb3
FTKX_MASK_WPB=6 run wpb6
FTKX_MASK_WPB=8 run wpb8
FTKX_MASK_WPB=12 run wpb12
FTKX_MASK_WPB=6 FTKX_MASK_ZCHUNK=16 run wpb6z16
