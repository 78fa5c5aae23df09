run() { name=$1; shift
  env "$@" python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$name', 'ms %.3f' % j['ms_per_step'], 'cull %.3f' % j['roofline']['kernel_ms_per_pass']['cull_kernel'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'])
"; }
run default A=1
run zc2 FTKX_CULL_ZC=2
run zc8 FTKX_CULL_ZC=8
run sc4 FTKX_CULL_STEP_CHUNK=4
run sc8 FTKX_CULL_STEP_CHUNK=8
run sc32 FTKX_CULL_STEP_CHUNK=32
run zc2sc8 FTKX_CULL_ZC=2 FTKX_CULL_STEP_CHUNK=8
run zc2sc4 FTKX_CULL_ZC=2 FTKX_CULL_STEP_CHUNK=4
