python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -2
for c in c3 c4 c2; do python tools/tracker_api_bench.py $c 2>/dev/null | tail -1; done
for c in c3 c4; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$c', 'ms %.3f' % j['ms_per_step'], 'e2e frac %.3f' % j['roofline_end_to_end']['frac'], j['wall_breakdown_ms_per_pass'])
"; done
