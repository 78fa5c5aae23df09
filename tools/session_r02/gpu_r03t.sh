timeout 900 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "generations or one_pass or big_vert or fused" 2>&1 | tail -5
python tools/ab_mask.py c4 4 -- "V=5" "V=6 TILE=3 PD=2" "V=6 TILE=3 PD=3" "V=6 TILE=3 PD=4" "V=6 TILE=0 PD=2" "V=6 TILE=0 PD=3" "V=6 TILE=0 PD=4" "V=6 TILE=9 PD=2" "V=6 TILE=7 PD=2" "V=6 TILE=7 PD=3" "V=6 TILE=1 PD=3" "V=6 TILE=8 PD=2"
