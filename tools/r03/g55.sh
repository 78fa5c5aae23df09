#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g55; rm -rf $O; mkdir -p $O
for w in 0 2 3 4 6 8 12; do
if [ $w = 0 ]; then unset FTKX_MASK_WPB; else export FTKX_MASK_WPB=$w; fi
python3 bench.py --config c2 --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c2_$w.json 2> $O/c2_$w.err; tail -1 $O/c2_$w.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('wpb $w', round(j['ms_per_step'],4), j['roofline_end_to_end']['frac'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['check'])" || tail -3 $O/c2_$w.err
done
