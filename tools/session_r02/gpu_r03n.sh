python tools/ab_mask.py c4 4 -- "TILE=0" "TILE=8" "TILE=9" "TILE=10" "TILE=9 YG=3" "TILE=9 YG=5"
python tools/ab_mask.py c3 6 -- "TILE=0 ZCHUNK=32" "TILE=8 ZCHUNK=32" "TILE=9 ZCHUNK=32" "TILE=10 ZCHUNK=32"
