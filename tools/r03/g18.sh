#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_trace.py tests/test_io_formats.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
bash tools/r03/g17.sh 2>&1 | grep -E "^host|^ctx"
for c in c2; do python3 tools/tracker_api_bench.py $c 2>&1 | tail -1; done
