timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations or one_pass or big_vert or fused" 2>&1 | tail -5
python tools/ab_mask.py c4 5 -- "V=5" "V=6" "V=6 SWIZZLE=136" "V=6 SWIZZLE=72"
python tools/ab_mask.py c3 6 -- "V=5" "V=6" "V=6 SWIZZLE=136"
