#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_g07; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_gpu_series.py tests/test_gpu_properties.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for c in c2 c5; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/trace_$c.log 2>&1 &&
  python3 tools/pass_timeline.py $O/trace_$c --first series_begin_kernel > $O/timeline_$c.txt 2>&1
  for w in 4 6 8; do
    FTKX_EXACT_WG_PER_CU=$w python3 bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-events > $O/${c}_wg$w.json 2>&1; echo "$c exact wg/cu $w: $(tail -1 $O/${c}_wg$w.json | cut -c1-110)"
  done
done
find $O -name "*.csv" -size +2M -delete
