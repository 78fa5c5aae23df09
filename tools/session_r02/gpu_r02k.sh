python -m pytest tests/test_shim.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | tail -8
