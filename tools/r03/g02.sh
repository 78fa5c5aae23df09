#!/bin/bash
# round 3, call 2: the series pass -- its own tests, then a smoke of the old paths after the file split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g02; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_series.py -q > $O/series.log 2>&1; echo "series rc=$?"; tail -15 $O/series.log
