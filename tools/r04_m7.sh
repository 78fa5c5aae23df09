#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/m7; mkdir -p $O
timeout -k 10 300 python tools/ab_mask.py c4 5 -- "ORDER=0 ZCHUNK=32" "ORDER=0 ZCHUNK=24" "ORDER=0 ZCHUNK=27" "ORDER=0 ZCHUNK=30" "ORDER=0 ZCHUNK=33" "ORDER=0 ZCHUNK=36" "ORDER=0 ZCHUNK=42" "ORDER=0 ZCHUNK=48" "ORDER=0 LCAP=24 LMIN=6" "ORDER=0 LCAP=30 LMIN=6" "ORDER=0 LCAP=33 LMIN=9" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab8.log
timeout -k 10 300 python tools/ab_mask.py c3 6 -- "ORDER=1 ZCHUNK=32" "ORDER=1 ZCHUNK=24" "ORDER=1 ZCHUNK=30" "ORDER=1 ZCHUNK=33" "ORDER=1 LCAP=24 LMIN=6" "ORDER=1 LCAP=30 LMIN=6" "ORDER=1 LCAP=33 LMIN=9" "ORDER=1 LCAP=24 LMIN=9" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab8.log
timeout -k 10 300 python tools/ab_mask.py c4s 6 -- "ZCHUNK=32" "ZCHUNK=24" "ZCHUNK=30" "ZCHUNK=33" "LCAP=24 LMIN=6" "LCAP=30 LMIN=6" "LCAP=33 LMIN=9" "LCAP=24 LMIN=9" "LCAP=18 LMIN=6" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab8.log
