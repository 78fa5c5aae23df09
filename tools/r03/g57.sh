#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g57; rm -rf $O; mkdir -p $O
run() { python3 bench.py --config $1 --steps 40 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker $2 > $O/x.json 2> $O/x.err; tail -1 $O/x.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1 $2', round(j['ms_per_step'],4))" || tail -3 $O/x.err; }
for cfg in c2 c5 c3; do
run $cfg ""
run $cfg "--no-kernel-events"
run $cfg ""
run $cfg "--no-kernel-events"
done
