timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations" 2>&1 | tail -5
python tools/ab_mask.py c4 4 -- "V=5" "V=6" "V=6 TILE=1" "V=6 TILE=2" "V=6 TILE=3" "V=6 TILE=4" "V=6 TILE=5" "V=6 PD=3"
