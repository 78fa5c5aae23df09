#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g12; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_series.py -x -q > $O/tests.log 2>&1; echo "rc=$?"; tail -12 $O/tests.log
for c in c2 c5; do
  for n in 1 2 3 4; do
    FTKX_SERIES_CHUNKS=$n python3 bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker --no-kernel-events > $O/${c}_ch$n.json 2>$O/${c}_ch$n.err; echo "$c chunks $n: $(tail -1 $O/${c}_ch$n.json | cut -c1-100)"
  done
  python3 bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/${c}_auto.json 2>$O/${c}_auto.err; tail -1 $O/${c}_auto.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$c auto', j['ms_per_step'], j['roofline']['frac'], j['roofline_end_to_end']['frac'], j['config']['pass'])"
done
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c5 -- python3 bench.py --config c5 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/trace_c5.log 2>&1 && python3 tools/pass_timeline.py $O/trace_c5 --first series_begin_kernel > $O/timeline_c5.txt 2>&1
find $O -name "*.csv" -size +2M -delete
