run() { python bench.py --config ${CFG:-c4} --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms/step=%.3f'%j['ms_per_step'], {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items()}, j['check']['hits'])"; }
run pd1
FTKX_MASK_PD=2 run pd2
FTKX_MASK_PD=3 run pd3
FTKX_MASK_PD=2 FTKX_MASK_SWIZZLE=16 run pd2_ntpriv
FTKX_MASK_PD=3 FTKX_MASK_SWIZZLE=16 run pd3_ntpriv
FTKX_MASK_PD=2 FTKX_MASK_SWIZZLE=8 run pd2_grp
FTKX_MASK_PD=3 FTKX_MASK_SWIZZLE=24 run pd3_ntpriv_grp
FTKX_MASK_PD=3 FTKX_MASK_ZCHUNK=64 run pd3_z64
FTKX_MASK_PD=3 FTKX_MASK_WPB=2 run pd3_wpb2
FTKX_MASK_PD=3 FTKX_MASK_WPB=8 run pd3_wpb8
