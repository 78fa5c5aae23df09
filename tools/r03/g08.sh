#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g08; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_series.py tests/test_shim.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print(j['ms_per_step'], j['roofline']['frac'], j['roofline_end_to_end']['frac'])
for k,v in j['configs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','frac','end_to_end_frac','hits','series_paths','error')})
print(json.dumps(j.get('streaming_tracker'), indent=1))
"; tail -3 $O/bench_c4.err
FTKX_UPLOAD=0 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_c4_oldupload.json 2> $O/bench_c4_oldupload.err; tail -1 $O/bench_c4_oldupload.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('old upload path:', json.dumps(j.get('streaming_tracker',{}).get('host_fed')))"
for c in c3 c2; do python3 tools/tracker_api_bench.py $c 2>&1 | tail -1; done
