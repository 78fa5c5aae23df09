#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/loop; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "slab_count" > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
python3 tools/slab_loopback.py > $O/plain.log 2>&1
python3 tools/slab_loopback.py --no-side > $O/noside.log 2>&1
python3 tools/slab_loopback.py --no-halo > $O/nohalo.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/slab_loopback.py --passes 4 > $O/traced.log 2>&1
python3 tools/pass_timeline.py $O/trace --first series_begin_kernel --skip 5 > $O/timeline_single.txt 2>&1
python3 tools/pass_timeline.py $O/trace --first series_begin_kernel --skip 9 > $O/timeline_pipelined.txt 2>&1
find $O -name "*kernel_trace.csv" -size +3M -delete
tail -3 $O/tests.log; cat $O/plain.log $O/noside.log $O/nohalo.log; cat $O/timeline_single.txt
