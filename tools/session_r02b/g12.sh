cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
python tools/ab_mask.py c4 6 -- "TILE=0" "YG=8" "YG=16" "YG=16 ZCHUNK=32" "YG=8 ZCHUNK=32" "ZCHUNK=32" > gpurun_out/g12/ab_c4.log 2>&1
cat gpurun_out/g12/ab_c4.log
python tools/ab_mask.py c3 8 -- "TILE=0" "YG=8" "YG=16" > gpurun_out/g12/ab_c3.log 2>&1
cat gpurun_out/g12/ab_c3.log
