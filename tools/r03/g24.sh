#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g24; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -x -q  > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
python3 bench.py --config c3 --exact-only --steps 3 --warmup 1 --no-cpu-baseline > $O/c3_exact.json 2> $O/c3_exact.err; tail -1 $O/c3_exact.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('c3 exact-only', j['ms_per_step'], j['value'], j['roofline']['kernel_ms_per_pass'])" || tail -3 $O/c3_exact.err
python3 bench.py --config c3o --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-streaming-tracker > $O/c3o.json 2> $O/c3o.err; tail -1 $O/c3o.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('c3o', j['ms_per_step'], j['value'], j['roofline']['kernel_ms_per_pass'], j['check'], j['config']['pass'])"
