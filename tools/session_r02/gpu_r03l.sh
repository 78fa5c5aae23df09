python tools/ab_mask.py c3 6 -- "ZCHUNK=16" "ZCHUNK=32" "ZCHUNK=64" "ZCHUNK=128" "ZCHUNK=256"
python tools/ab_mask.py c4 4 -- "ZCHUNK=32" "ZCHUNK=64" "ZCHUNK=128"
