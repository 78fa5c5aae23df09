mkdir -p gpurun_out/r02f
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config ${CFG:-c4} --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${CFG:-c4}', '$name', 'ms/pass %.3f' % j['ms_per_step'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'], j['roofline']['kernel'], 'hits', j['check']['hits'])
" | tee -a gpurun_out/r02f/variants.txt
}
run default A=1
for yg in 2 4 8 16; do run yg$yg FTKX_MASK_YG=$yg; done
run zc16 FTKX_MASK_ZCHUNK=16
run zc64 FTKX_MASK_ZCHUNK=64
run zc128 FTKX_MASK_ZCHUNK=128
run nt FTKX_MASK_SWIZZLE=12
run ntpriv FTKX_MASK_SWIZZLE=24
run sw0 FTKX_MASK_SWIZZLE=0
run sw1 FTKX_MASK_SWIZZLE=1
run yg4zc64 FTKX_MASK_YG=4 FTKX_MASK_ZCHUNK=64
