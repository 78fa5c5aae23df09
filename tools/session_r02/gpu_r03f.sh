python -m pytest tests/test_gpu_parity.py tests/test_io_formats.py tests/test_shim.py -m gpu -x -q 2>&1 | tail -2
bash tools/session_r02/gpu_r03e.sh 2>&1 | tail -5
