#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g48; rm -rf $O; mkdir -p $O
for w in 4 8 16 32 64 256; do
FTKX_COPY_WGS=$w python3 bench.py --config c2 --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c2_$w.json 2> $O/c2_$w.err; tail -1 $O/c2_$w.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('wgs $w', round(j['ms_per_step'],4), 'latency', round(j['single_pass_latency_ms'],4), j['roofline_end_to_end']['frac'], j['roofline']['avg_launch_ms'])" || tail -3 $O/c2_$w.err
done
FTKX_SERIES_COPY=0 python3 bench.py --config c2 --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c2_nocopy.json 2> $O/c2_nocopy.err; tail -1 $O/c2_nocopy.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('no copy', round(j['ms_per_step'],4), 'latency', round(j['single_pass_latency_ms'],4), j['roofline_end_to_end']['frac'], j['roofline']['avg_launch_ms'])"
