#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g16; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_trace.py tests/test_gpu_parity.py tests/test_io_formats.py tests/test_shim.py tests/test_gpu_multirank.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?"; tail -5 $O/tests.log
for c in c2 c3; do python3 tools/tracker_api_bench.py $c 2>&1 | tail -1; done
python3 bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c2.json 2> $O/c2.err; tail -1 $O/c2.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['pass2']))"
python3 bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c5.json 2> $O/c5.err; tail -1 $O/c5.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['pass2']))"
