#!/bin/bash
# The whole `-m gpu` suite in one process on the GPU box, log under gpurun_out/:   gpurun --timeout 1190 -- 'bash tools/gpu_suite.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/gpu_suite; rm -rf $O; mkdir -p $O
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q -rs --durations=15 > $O/gpu_tests.log 2>&1; echo "rc=$?"; tail -25 $O/gpu_tests.log | cut -c1-300
