#!/bin/bash
# profiles of all four configs + the exact-only line
for cfg in c4 c3 c2 c5; do bash tools/collect_profiles.sh $cfg r03 | tail -1 | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r03_c3_exact_only; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config c3 --exact-only --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_trace.log 2>&1
python3 bench.py --config c3 --exact-only --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
find $O -name "*kernel_trace.csv" -size +3M -delete
tail -1 $O/bench_plain.json | cut -c1-200
