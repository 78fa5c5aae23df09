#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g53; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -1 $O/bench_default.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(j['metric'], j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_end_to_end']['frac'], j['single_pass_latency_ms'])
for k,v in j['configs'].items(): print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','frac','end_to_end_frac','hits','error')})
print('streaming', json.dumps(j.get('streaming_tracker'))[:900])
print('cpu', j['cpu_baseline'])
print('pass2', j['pass2'])
" || tail -5 $O/bench_default.err
