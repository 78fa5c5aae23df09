python tools/ab_mask.py c4 4 -- "V=5" "V=6 TILE=3" "V=6 TILE=0" "V=6 TILE=9" "V=6 TILE=3 PD=3" "V=6 TILE=3 PD=1"
