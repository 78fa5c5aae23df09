#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace + separate PMC passes for one bench config, plus the FETCH_SIZE calibration and
# the timeline of one pass.   usage: bash tools/collect_profiles.sh <config> <tag>
CFG=${1:-c4}; TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_${TAG}_${CFG}; rm -rf $OUT; mkdir -p $OUT
LIGHT="--no-cpu-baseline --no-other-configs --no-streaming-tracker --no-sustained"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $CFG --steps 5 --warmup 2 $LIGHT > $OUT/bench_under_trace.log 2>&1
# (bench.py's passes under the trace: 3 warm-up + 5 timed with two in flight, then 3 on their own (latency) and 3 with events around every kernel:
# --skip 7 = a timed pass with another one in flight; --skip 4 = a pass on its own)
python3 tools/pass_timeline.py $OUT/trace --first series_begin_kernel,series_one_kernel --skip 7 > $OUT/timeline.txt 2>&1
python3 tools/pass_timeline.py $OUT/trace --first series_begin_kernel,series_one_kernel --skip 4 > $OUT/timeline_single.txt 2>&1
# the same passes with NO events between the kernels (what `sustained`, the side configurations and the streaming tracker run): the split pass's
# tail next to the next pass's mask kernel (3 warm-up + 8 timed + 3 latency passes: --skip 6 = a timed pass)
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 bench.py --config $CFG --steps 8 --warmup 2 --no-kernel-events $LIGHT > $OUT/bench_under_trace_plain.log 2>&1
python3 tools/pass_timeline.py $OUT/trace_plain --first series_begin_kernel,series_one_kernel --skip 6 > $OUT/timeline_overlap.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --config $CFG --steps 2 --warmup 1 $LIGHT > $OUT/bench_under_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --config $CFG --steps 2 --warmup 1 $LIGHT > $OUT/bench_under_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- python3 tools/calibrate_fetch.py > $OUT/calib.log 2>&1
python3 bench.py --config $CFG --steps 20 --warmup 3 $LIGHT > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*kernel_trace.csv" -size +3M -delete
tail -1 $OUT/bench_plain.json | cut -c1-300
