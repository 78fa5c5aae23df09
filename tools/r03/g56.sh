#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g56; rm -rf $O; mkdir -p $O
run() { python3 bench.py --config $1 --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker $2 > $O/x.json 2> $O/x.err; tail -1 $O/x.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1 $2 $3', round(j['ms_per_step'],4), round(j['roofline_end_to_end']['frac'],4), j['check'].get('hits'))" || tail -3 $O/x.err; }
for cfg in c2 c5 c3; do
run $cfg "" events
run $cfg "--no-kernel-events" noevents
done
FTKX_TWO_LEVEL=0 run c2 "" two_level_off
FTKX_TWO_LEVEL=0 run c5 "" two_level_off
