python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python tools/ab_mask.py c2 10 -- "PD=1" 2>&1 | grep -v amdgpu.ids
for c in c2 c3 c4 c5; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$c', 'ms %.3f' % j['ms_per_step'], 'e2e frac %.3f' % j['roofline_end_to_end']['frac'], j['roofline']['kernel_ms_per_pass'])
"; done
