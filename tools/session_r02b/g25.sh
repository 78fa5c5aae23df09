cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g25
python tools/ab_mask.py c4 6 -- "TILE=0" "TAIL=1,8" "TAIL=2,8" "TAIL=4,8" "TAIL=2,16" "ZCHUNK=64 TAIL=2,16" "ZCHUNK=64 TAIL=4,8" > gpurun_out/g25/ab_c4.log 2>&1
cat gpurun_out/g25/ab_c4.log
python tools/ab_mask.py c3 8 -- "TILE=0" "TAIL=1,8" "TAIL=2,8" "TAIL=4,8" "TAIL=2,16" > gpurun_out/g25/ab_c3.log 2>&1
cat gpurun_out/g25/ab_c3.log
