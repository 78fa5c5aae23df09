#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g49; rm -rf $O; mkdir -p $O
for b in 16 15 14 13 12; do
for cfg in c2 c5; do
FTKX_SERIES_BINS_LOG2=$b python3 bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/${cfg}_$b.json 2> $O/${cfg}_$b.err; tail -1 $O/${cfg}_$b.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$cfg bins $b', round(j['ms_per_step'],4), 'latency', round(j['single_pass_latency_ms'],4), j['roofline_end_to_end']['frac'], j['roofline']['avg_launch_ms'], j['config']['pass'][-30:])" || tail -3 $O/${cfg}_$b.err
done
done
