python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in c2 c3 c4 c5; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$c', 'ms %.3f' % j['ms_per_step'], 'e2e frac %.3f' % j['roofline_end_to_end']['frac'], j['roofline']['kernel_ms_per_pass'], j['wall_breakdown_ms_per_pass'])
"; FTKX_NO_SPECULATION=1 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$c nospec', 'ms %.3f' % j['ms_per_step'])
"; done
