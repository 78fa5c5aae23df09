#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g36; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_series.py -x -q -k short_chain > $O/tests.log 2>&1; echo "rc=$?"; grep -n "^E " $O/tests.log | head
