cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g11
python tools/ab_mask.py c4 5 -- "YG=8" "YG=8 ZCHUNK=32" "YG=16 ZCHUNK=32" "YG=32" "YG=32 ZCHUNK=32" "YG=8 PD=3" "YG=8 ZCHUNK=16" "YG=6" > gpurun_out/g11/ab_c4.log 2>&1
cat gpurun_out/g11/ab_c4.log
python tools/ab_mask.py c3 8 -- "YG=8" "YG=16" "YG=8 ZCHUNK=16" "YG=16 ZCHUNK=64" "YG=4" > gpurun_out/g11/ab_c3.log 2>&1
cat gpurun_out/g11/ab_c3.log
