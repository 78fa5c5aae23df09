python tools/ab_mask.py c4 6 -- "TILE=0" "TILE=0 SWIZZLE=40" "TILE=1" "TILE=1 SWIZZLE=40" "TILE=6 PD=3" "TILE=6 PD=3 SWIZZLE=40" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c4_prio.txt
python tools/ab_mask.py c3 10 -- "TILE=0" "TILE=0 SWIZZLE=40" "TILE=6 PD=3" "TILE=0 ZCHUNK=16" "TILE=0 ZCHUNK=64" "TILE=6 PD=3 ZCHUNK=64" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c3.txt
