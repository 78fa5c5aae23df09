#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g04; rm -rf $O; mkdir -p $O
FTKX_SERIES_DEBUG=1 python3 bench.py --config c3 --steps 12 --warmup 2 --no-cpu-baseline > $O/c3.json 2> $O/c3.err
cat $O/c3.err | head -70
