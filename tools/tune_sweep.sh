python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --config ${CFG:-c4} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'])"; }
run default
FTKX_MASK_EDGE=0 run halo
FTKX_MASK_EDGE=1 FTKX_MASK_SWIZZLE=0 run edge_noswz
FTKX_MASK_EDGE=1 FTKX_MASK_ZCHUNK=32 run edge_z32
CFG=c3 run c3
CFG=c3 FTKX_MASK_EDGE=0 run c3halo
CFG=c2 run c2
CFG=c1 run c1
