run() { python bench.py --config ${CFG:-c4} --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'])"; }
run pd1
FTKX_MASK_SWIZZLE=10 run pd1_nostore
FTKX_MASK_PD=3 FTKX_MASK_SWIZZLE=10 run pd3_nostore
FTKX_TWO_LEVEL=0 run no_summary
FTKX_MASK_WPB=8 run wpb8
FTKX_MASK_WPB=2 run wpb2
FTKX_MASK_ZCHUNK=64 run z64
FTKX_MASK_ZCHUNK=16 run z16
