cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g10
python tools/ab_mask.py c4 5 -- "TILE=0" "ZCHUNK=128" "ZCHUNK=32" "YG=8" "YG=2" "YG=16" "ZCHUNK=128 YG=8" > gpurun_out/g10/ab_c4.log 2>&1
cat gpurun_out/g10/ab_c4.log
python tools/ab_mask.py c3 8 -- "TILE=0" "ZCHUNK=64" "ZCHUNK=16" "YG=8" "YG=2" > gpurun_out/g10/ab_c3.log 2>&1
cat gpurun_out/g10/ab_c3.log
