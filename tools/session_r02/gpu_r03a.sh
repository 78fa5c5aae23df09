FTKX_TRACE_PROF=1 python bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "trace:" | tail -3
FTKX_TRACE_PROF=1 FTKX_TRACE_THREADS=1 python bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "trace:" | tail -1
FTKX_TRACE_PROF=1 FTKX_TRACE_THREADS=64 python bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "trace:" | tail -1
python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -2
