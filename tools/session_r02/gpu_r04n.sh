python3 tools/ab_mask.py c4 4 -- "V=6" "V=6 SWIZZLE=24" "V=5" "V=5 SWIZZLE=24"
python3 tools/ab_mask.py c3 5 -- "V=6" "V=6 SWIZZLE=24"
