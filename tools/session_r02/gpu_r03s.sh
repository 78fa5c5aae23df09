timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations or one_pass or big_vert or fused" 2>&1 | tail -5
python tools/ab_mask.py c4 4 -- "V=5" "V=6 TILE=3" "V=6 TILE=0" "V=6 TILE=9" "V=6 TILE=3 ZCHUNK=64" "V=6 TILE=1"
