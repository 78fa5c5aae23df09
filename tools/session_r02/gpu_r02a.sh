set -x
mkdir -p gpurun_out/r02a
( time python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "fullsize or c4 or c2 or c5 or wraps" ) > gpurun_out/r02a/tests_new.log 2>&1
tail -5 gpurun_out/r02a/tests_new.log
for c in c2 c3 c5 c4; do python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_$c.json 2> gpurun_out/r02a/bench_$c.err; tail -c 1500 gpurun_out/r02a/bench_$c.json; done
python bench.py --config c3 --exact-only --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_c3_exact.json 2>&1; tail -c 1200 gpurun_out/r02a/bench_c3_exact.json
nproc; free -g | head -2
