mkdir -p gpurun_out/r02e
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config ${CFG:-c4} --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${CFG:-c4}', '$name', 'ms/pass %.3f' % j['ms_per_step'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'], j['roofline']['kernel'], 'hits', j['check']['hits'])
" | tee -a gpurun_out/r02e/variants.txt
}
run default A=1
run pd1 FTKX_MASK_PD=1
run pd3 FTKX_MASK_PD=3
for t in 1 2 3 5 6 7; do run tile$t FTKX_MASK_TILE=$t; done
run tile3pd3 FTKX_MASK_TILE=3 FTKX_MASK_PD=3
run tile5pd3 FTKX_MASK_TILE=5 FTKX_MASK_PD=3
run tile6pd3 FTKX_MASK_TILE=6 FTKX_MASK_PD=3
run v4 FTKX_MASK_V=4
CFG=c3 run default A=1
CFG=c3 run tile5 FTKX_MASK_TILE=5
CFG=c3 run tile6 FTKX_MASK_TILE=6
python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -5
