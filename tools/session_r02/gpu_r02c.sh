mkdir -p gpurun_out/r02c
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$name', 'ms/pass %.3f' % j['ms_per_step'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'], j['roofline']['kernel'], 'hits', j['check']['hits'])
" | tee -a gpurun_out/r02c/variants.txt
}
run default A=1
run pd1 FTKX_MASK_PD=1
run pd3 FTKX_MASK_PD=3
run tile5 FTKX_MASK_TILE=5
run tile5pd3 FTKX_MASK_TILE=5 FTKX_MASK_PD=3
run tile8 FTKX_MASK_TILE=8
run tile4 FTKX_MASK_TILE=4
run tile6 FTKX_MASK_TILE=6
run tile7 FTKX_MASK_TILE=7
run tile9 FTKX_MASK_TILE=9
run tile10 FTKX_MASK_TILE=10
run v4 FTKX_MASK_V=4
run v4pd2 FTKX_MASK_V=4 FTKX_MASK_PD=2
run v4ry8 FTKX_MASK_V=4 FTKX_MASK_RY=8
