python tools/ab_mask.py c4 6 -- "TILE=0" "TILE=1" "TILE=6 PD=3" "TILE=3" "TILE=5" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c4_lean.txt
