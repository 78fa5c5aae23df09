python tools/ab_mask.py c4 8 -- "TILE=0" "TILE=0 SWIZZLE=72" "TILE=6 PD=3" "TILE=6 PD=3 SWIZZLE=72" "TILE=1" "TILE=1 SWIZZLE=72" "TILE=0 PD=1" "TILE=0 PD=1 SWIZZLE=72" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c4_early2.txt
python tools/ab_mask.py c3 10 -- "TILE=0" "TILE=0 SWIZZLE=72" 2>&1 | grep -v amdgpu.ids
