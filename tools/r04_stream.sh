#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/stream; rm -rf $O; mkdir -p $O
python3 tools/tracker_step_breakdown.py c4 > $O/plain.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/tracker_step_breakdown.py c4 > $O/traced.log 2>&1
python3 tools/pass_timeline.py $O/trace --first series_begin_kernel --skip 3 > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -size +3M -delete
grep -v amdgpu $O/plain.log; cat $O/timeline.txt
