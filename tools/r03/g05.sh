#!/bin/bash
# full GPU suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g05; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -rs > $O/gpu_tests.log 2>&1; echo "rc=$?"; tail -15 $O/gpu_tests.log
