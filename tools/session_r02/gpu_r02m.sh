mkdir -p gpurun_out/r02m
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r02m/tests.log 2>&1; tail -6 gpurun_out/r02m/tests.log
for c in c2 c5 c3 c4; do python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/r02m/bench_$c.json; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r02m/bench_$c.json") if l.startswith("{")][-1])
print("$c", "ms %.3f" % j["ms_per_step"], j["roofline"]["kernel_ms_per_pass"], j["wall_breakdown_ms_per_pass"], j["pass2"])
PY
done
python bench.py --config c3 --exact-only --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/r02m/bench_c3_exact.json; tail -c 900 gpurun_out/r02m/bench_c3_exact.json; echo
python bench.py --config c3o --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02m/c3o.err > gpurun_out/r02m/bench_c3o.json; tail -c 1200 gpurun_out/r02m/bench_c3o.json; tail -3 gpurun_out/r02m/c3o.err
