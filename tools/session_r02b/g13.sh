cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g13
python tools/ab_mask.py c4 6 -- "YG=8" "YG=8 PD=3" "YG=8 SWIZZLE=72" "YG=8 PD=3 SWIZZLE=72" "YG=16 ZCHUNK=32" "YG=16 ZCHUNK=32 PD=3" > gpurun_out/g13/ab_c4.log 2>&1
cat gpurun_out/g13/ab_c4.log
python tools/ab_mask.py c2 10 -- "YG=4" "YG=8" "YG=16" "YG=2" > gpurun_out/g13/ab_c2.log 2>&1
cat gpurun_out/g13/ab_c2.log
