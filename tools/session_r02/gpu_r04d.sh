timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations or one_pass or big_vert or fused" 2>&1 | tail -5
python tools/ab_mask.py c4 5 -- "V=5" "V=6 TILE=3 PD=3 SWIZZLE=40" "V=6 TILE=3 PD=3 SWIZZLE=104" "V=6 TILE=3 PD=2 SWIZZLE=40" "V=6 TILE=3 PD=2 SWIZZLE=104" "V=6 TILE=3 PD=4 SWIZZLE=104" "V=6 TILE=3 PD=4 SWIZZLE=40"
