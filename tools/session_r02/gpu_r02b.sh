set -x
mkdir -p gpurun_out/r02b
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r02b/tests.log 2>&1
tail -30 gpurun_out/r02b/tests.log
for c in c4 c3 c2 c5; do python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02b/bench_$c.json 2> gpurun_out/r02b/bench_$c.err; tail -c 2500 gpurun_out/r02b/bench_$c.json; tail -3 gpurun_out/r02b/bench_$c.err; done
