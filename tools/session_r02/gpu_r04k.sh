timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python tools/ab_mask.py c4 5 -- "V=5" "V=6" "V=6 TILE=2"
python tools/ab_mask.py c3 6 -- "V=5" "V=6" "V=6 TILE=2"
