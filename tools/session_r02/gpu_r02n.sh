mkdir -p gpurun_out/r02n
python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for c in c2 c5 c3; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r02n/bench_$c.json; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r02n/bench_$c.json") if l.startswith("{")][-1])
print("$c", "ms %.3f" % j["ms_per_step"], j["roofline"]["kernel_ms_per_pass"], j["wall_breakdown_ms_per_pass"], {k: j["pass2"][k] for k in ("records","curves","trace_ms","post_process_ms")})
PY
done
