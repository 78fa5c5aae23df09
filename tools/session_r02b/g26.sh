cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g26
python tools/ab_mask.py c2 10 -- "WPB=4" "WPB=2" "WPB=3" "WPB=6" "WPB=8" "WPB=12" "WPB=4 SWIZZLE=0" "WPB=4 SWIZZLE=1" > gpurun_out/g26/ab_c2.log 2>&1
cat gpurun_out/g26/ab_c2.log
