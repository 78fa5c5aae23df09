cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g20
python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/g20/pytest.log 2>&1; echo "pytest rc $?"
tail -40 gpurun_out/g20/pytest.log | cut -c1-400
