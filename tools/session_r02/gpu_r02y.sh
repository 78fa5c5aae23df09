python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "streaming or multi_device" 2>&1 | tail -5
