cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g16
python tools/ab_mask.py c5 8 -- "FTKX_VEC_BX=2048" "FTKX_VEC_BX=1024" "FTKX_VEC_BX=512" "FTKX_VEC_BX=256" "FTKX_VEC_BX=128" > gpurun_out/g16/ab_c5.log 2>&1
cat gpurun_out/g16/ab_c5.log
