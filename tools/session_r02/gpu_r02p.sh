python tools/ab_mask.py c4 6 -- "TILE=0" "TILE=0 SWIZZLE=40" "TILE=1" "TILE=1 SWIZZLE=40" "TILE=6 PD=3" "TILE=6 PD=3 SWIZZLE=40" "TILE=3" "TILE=5" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c4_fast.txt
python -m pytest tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
