cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g02
python tools/ab_mask.py c4 5 -- "TILE=0" "PD=2 SWIZZLE=72" "PD=2" "PD=4" "PD=3 SWIZZLE=72" "PD=2 SWIZZLE=72 ZCHUNK=128" > gpurun_out/g02/ab_c4.log 2>&1
cat gpurun_out/g02/ab_c4.log
python tools/ab_mask.py c3 8 -- "TILE=0" "PD=2 SWIZZLE=72" "PD=2" "ZCHUNK=32" "PD=2 SWIZZLE=72 ZCHUNK=32" > gpurun_out/g02/ab_c3.log 2>&1
cat gpurun_out/g02/ab_c3.log
