#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g13; rm -rf $O; mkdir -p $O
for c in c5 c2; do
FTKX_SERIES_CHUNKS=2 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming-tracker --no-kernel-events > $O/trace_$c.log 2>&1 && python3 tools/pass_timeline.py $O/trace_$c --first fetch_desc_kernel > $O/timeline_$c.txt 2>&1
done
find $O -name "*.csv" -size +2M -delete
