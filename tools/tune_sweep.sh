run() { python bench.py --config ${CFG:-c4} --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', {k:round(v,3) for k,v in j['roofline']['kernel_ms_per_pass'].items() if k=='mask_kernel'})"; }
run ry4_pd1
FTKX_MASK_SWIZZLE=186 FTKX_MASK_RY=8 FTKX_MASK_YG=4 run ry8_ntpriv_yg4_noedge_loadsonly
FTKX_MASK_SWIZZLE=186 FTKX_MASK_RY=8 FTKX_MASK_YG=4 FTKX_MASK_PD=0 run ry8_ntpriv_yg4_noedge_loadsonly_pd0
FTKX_MASK_SWIZZLE=154 FTKX_MASK_RY=8 FTKX_MASK_YG=4 run ry8_ntpriv_yg4_edge_loadsonly
FTKX_MASK_SWIZZLE=170 FTKX_MASK_YG=4 run ry4_yg4_noedge_loadsonly
FTKX_MASK_SWIZZLE=170 FTKX_MASK_YG=4 FTKX_MASK_PD=0 run ry4_yg4_noedge_loadsonly_pd0
FTKX_MASK_SWIZZLE=138 FTKX_MASK_YG=4 run ry4_yg4_edge_loadsonly
