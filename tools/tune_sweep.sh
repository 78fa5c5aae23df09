run() { python bench.py --config ${CFG:-c4} --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items() if k=='mask_kernel'}, j['check']['hits'])"; }
run v5_128x16_pd2
FTKX_MASK_TILE=10 run v5_128x24_c6r4
FTKX_MASK_TILE=11 run v5_128x24_c3r8
FTKX_MASK_TILE=10 FTKX_MASK_YG=2 run v5_128x24_c6r4_yg2
FTKX_MASK_TILE=9 run v5_128x20
run v5_128x16_pd2_again
