#!/bin/bash
# round 3, call 3: the series pass in bench.py -- per-config lines (device-driven vs host-driven) and a timeline of one pass each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g03; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_gpu_series.py -q -rs > $O/series.log 2>&1; tail -12 $O/series.log
for c in c2 c3 c5; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/trace_$c.log 2>&1 &&
  python3 tools/pass_timeline.py $O/trace_$c --first series_begin_kernel > $O/timeline_$c.txt 2>&1
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --host-driven > $O/bench_${c}_host.json 2> $O/bench_${c}_host.err
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/bench_${c}_noev.json 2> $O/bench_${c}_noev.err
done
python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --host-driven > $O/bench_c4_host.json 2> $O/bench_c4_host.err
find $O -name "*.csv" -size +2M -delete
for c in c2 c2_host c2_noev c3 c3_host c3_noev c5 c5_host c5_noev c4 c4_host; do python3 - <<P
import json
try:
    j=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1])
    if "roofline" in j: print("$c", round(j["ms_per_step"],4), "kernel frac", round(j["roofline"]["frac"],3), "e2e", round(j["roofline_end_to_end"]["frac"],3), j["roofline"]["kernel_ms_per_pass"], j["config"].get("pass"), j["check"])
    else: print("$c", j)
except Exception as e: print("$c", "failed", e, open("$O/bench_$c.err").read()[-800:])
P
done
