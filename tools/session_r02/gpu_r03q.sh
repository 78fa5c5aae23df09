timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations" 2>&1 | tail -5
python tools/ab_mask.py c4 4 -- "V=5" "V=6 TILE=3" "V=6 TILE=7" "V=6 TILE=8" "V=6 TILE=9" "V=6 TILE=10" "V=6 TILE=11" "V=6 TILE=12" "V=6 TILE=7 ZCHUNK=64" "V=6 TILE=9 ZCHUNK=64"
