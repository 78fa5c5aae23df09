#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g27; rm -rf $O; mkdir -p $O
for fan in 2 1; do
FTKX_TILE_FAN=$fan python3 bench.py --config c3 --exact-only --steps 3 --warmup 1 --no-cpu-baseline > $O/c3_exact_$fan.json 2> $O/c3_exact_$fan.err; tail -1 $O/c3_exact_$fan.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('c3 exact-only fan $fan', j['ms_per_step'], j['value'], j['roofline']['kernel_ms_per_pass']['tile_kernel'], j['check'])" || tail -3 $O/c3_exact_$fan.err
done
