#!/bin/bash
# round 3, call 1: baseline of this round's box -- bench lines for c2/c3/c5/c4 and a kernel-trace timeline of one pass each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g01; rm -rf $O; mkdir -p $O
for c in c2 c3 c5; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/trace_$c.log 2>&1 &&
  python3 tools/pass_timeline.py $O/trace_$c > $O/timeline_$c.txt 2>&1
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  echo "$c done"; 
done
python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
find $O -name "*.csv" -size +2M -delete
for c in c2 c3 c5 c4; do python3 - <<P
import json
j=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1])
print("$c", round(j["ms_per_step"],4), round(j["roofline"]["frac"],3), round(j["roofline_end_to_end"]["frac"],3), j["roofline"]["kernel_ms_per_pass"], j["wall_breakdown_ms_per_pass"])
P
done
