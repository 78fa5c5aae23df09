python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { python bench.py --config ${CFG:-c4} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'])"; }
run default
FTKX_MASK_SWIZZLE=5 run nt
FTKX_MASK_ZCHUNK=128 run z128
FTKX_MASK_ZCHUNK=32 run z32
FTKX_MASK_ZCHUNK=128 FTKX_MASK_SWIZZLE=5 run z128nt
