#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g47; rm -rf $O; mkdir -p $O
LIGHT="--no-cpu-baseline --no-other-configs --no-streaming-tracker"
for cfg in c3 c2; do
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 2 $LIGHT > $O/bench_$cfg.log 2>&1
python3 tools/pass_timeline.py $O/trace_$cfg --first series_begin_kernel --skip 8 > $O/timeline_$cfg.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/timeline_$cfg.txt
done
