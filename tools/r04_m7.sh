#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/m7; mkdir -p $O
for cfg in c3 c4s c4n4; do
  timeout -k 10 300 python tools/ab_mask.py $cfg 8 -- "ZCHUNK=32" "LCAP=16 LMIN=4" "LCAP=20 LMIN=5" "LCAP=24 LMIN=6" "LCAP=24 LMIN=12" "LCAP=28 LMIN=7" "LCAP=24 LMIN=3" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab5.log
done
