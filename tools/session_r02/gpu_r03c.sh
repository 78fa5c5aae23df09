python -m pytest tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -2
python tools/ab_mask.py c4 6 -- "TILE=0" "TILE=0 PD=1" "TILE=0 PD=3" "TILE=1" "TILE=1 PD=1" "TILE=6 PD=3" "TILE=6 PD=2" "TILE=3" "TILE=5" "TILE=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_c4_early.txt
