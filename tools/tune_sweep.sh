run() { python bench.py --config ${CFG:-c4} --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items() if k=='mask_kernel'}, j['check']['hits'])"; }
export CFG=c3
run v5_default
FTKX_MASK_V=4 run v4
FTKX_MASK_ZCHUNK=16 run v5_z16
FTKX_MASK_ZCHUNK=64 run v5_z64
FTKX_MASK_PD=1 run v5_pd1
FTKX_MASK_PD=3 run v5_pd3
FTKX_MASK_TILE=8 run v5_128x12
FTKX_MASK_V=4 FTKX_MASK_ZCHUNK=16 run v4_z16
run v5_default_again
