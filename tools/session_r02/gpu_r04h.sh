timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations" 2>&1 | tail -3
python tools/ab_mask.py c4 5 -- "V=5" "V=6" "V=6 SWIZZLE=136"
python tools/ab_mask.py c3 6 -- "V=6" "V=6 SWIZZLE=136"
