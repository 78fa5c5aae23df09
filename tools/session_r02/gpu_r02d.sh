mkdir -p gpurun_out/r02d
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$name', 'ms/pass %.3f' % j['ms_per_step'], 'mask %.3f' % j['roofline']['kernel_ms_per_pass']['mask_kernel'], j['roofline']['kernel'], 'hits', j['check']['hits'])
" | tee -a gpurun_out/r02d/variants.txt
}
run default A=1
run tile5 FTKX_MASK_TILE=5
run tile5pd3 FTKX_MASK_TILE=5 FTKX_MASK_PD=3
run tile8 FTKX_MASK_TILE=8
run tile8pd3 FTKX_MASK_TILE=8 FTKX_MASK_PD=3
run tile4 FTKX_MASK_TILE=4
run tile10 FTKX_MASK_TILE=10
run v4 FTKX_MASK_V=4
python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -15
