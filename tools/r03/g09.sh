#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g09; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_properties.py -x -q > $O/tests.log 2>&1; echo "rc=$?"; tail -30 $O/tests.log
