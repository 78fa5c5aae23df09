run() { name=$1; cfg=$2; shift; shift
  env "$@" python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = j['roofline']['kernel_ms_per_pass']
print('$name', '$cfg', 'ms %.3f' % j['ms_per_step'], 'cull %.3f' % k['cull_kernel'], 'mask %.3f' % k['mask_kernel'], 'exact %.3f' % k['exact_kernel'], j['check']['hits'], j['stats'])
"; }
for cfg in c2 c5; do
for w in 2 4 8 16; do run wg$w $cfg FTKX_EXACT_WG_PER_CU=$w; done
done
run wg4 c3 FTKX_EXACT_WG_PER_CU=4
python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
