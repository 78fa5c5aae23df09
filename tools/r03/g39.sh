#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g50; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_series.py -x -q -k "pipelined or short" > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log; grep -n "^E " $O/tests.log | head -20
for cfg in c2 c5 c3 c4; do
python3 bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/$cfg.json 2> $O/$cfg.err; tail -1 $O/$cfg.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$cfg', round(j['ms_per_step'],4), 'latency', round(j['single_pass_latency_ms'],4), j['roofline']['frac'], j['roofline_end_to_end']['frac'], j['roofline']['avg_launch_ms'], j['check'].get('hits'), j['config']['pass'][-40:])" || tail -3 $O/$cfg.err
done
