#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g06; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_gpu_series.py -q -x > $O/series.log 2>&1; tail -3 $O/series.log
for c in c2 c5 c3; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/trace_$c.log 2>&1 &&
  python3 tools/pass_timeline.py $O/trace_$c --first series_begin_kernel > $O/timeline_$c.txt 2>&1
done
for m in 0; do
  FTKX_SERIES_STORE=$m python3 bench.py --config c2 --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-events > $O/c2_store$m.json 2>&1; echo "store $m: $(cat $O/c2_store$m.json | tail -1 | cut -c1-120)"
done
python3 bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c4_full.json 2> $O/bench_c4_full.err; tail -1 $O/bench_c4_full.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(j['ms_per_step'], j['roofline']['frac'], j['roofline_end_to_end']['frac'])
for k,v in j['configs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','frac','end_to_end_frac','hits','series_paths','error')})
"
find $O -name "*.csv" -size +2M -delete
