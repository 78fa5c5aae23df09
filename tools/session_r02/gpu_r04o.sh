timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for cfg in c4 c3; do python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = j['roofline']['kernel_ms_per_pass']
print('$cfg', 'ms %.3f' % j['ms_per_step'], 'frac %.3f' % j['roofline']['frac'], 'e2e %.3f' % j['roofline_end_to_end']['frac'], j['roofline']['kernel'], {a: round(b, 3) for a, b in k.items()}, j['check'], j['stats'])
"; done
FTKX_U_ROWS=1 python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k = j['roofline']['kernel_ms_per_pass']
print('c4 rows1', 'ms %.3f' % j['ms_per_step'], 'frac %.3f' % j['roofline']['frac'], {a: round(b, 3) for a, b in k.items()}, j['stats'])
"
