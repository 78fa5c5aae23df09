cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g17
python tools/ab_mask.py c5 8 -- "FTKX_VEC_BX=4096" "FTKX_VEC_BX=1024" "FTKX_VEC_BX=512" "FTKX_VEC_BX=256" > gpurun_out/g17/ab_c5.log 2>&1
cat gpurun_out/g17/ab_c5.log
python -m pytest tests -m gpu -x -q -k "gyre or vector or vec or fullsize or one_pass or cull_ahead or mask_kernel" > gpurun_out/g17/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/g17/pytest.log
