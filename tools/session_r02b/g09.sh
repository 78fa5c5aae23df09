cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g09
python -m pytest tests -m gpu -x -q > gpurun_out/g09/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/g09/pytest.log
python tools/tracker_api_bench.py > gpurun_out/g09/tracker_api.log 2>&1; tail -12 gpurun_out/g09/tracker_api.log
