timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multi_device" 2>&1 | tail -8
