python tools/ab_mask.py c4 4 -- "V=5" "V=6 TILE=3" "V=6 TILE=3 PD=3" "V=6 TILE=3 PD=1" "V=6 TILE=6" "V=6 TILE=6 PD=3" "V=6 TILE=3 YG=2" "V=6 TILE=3 YG=8" "V=6 TILE=3 YG=3" "V=6 TILE=3 ZCHUNK=64"
